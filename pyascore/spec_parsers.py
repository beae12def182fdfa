"""``from pyascore import spec_parsers`` (the reference's pyascore/parsing/spec_parsers.py): the
dependency-free readers of :mod:`pyascore_amd.ingest`."""
from pyascore_amd.ingest import MzMLExtractor, MzXMLExtractor, SpectraParser  # noqa: F401
