"""``from pyascore import id_parsers`` (the reference's pyascore/parsing/id_parsers.py): the
dependency-free readers of :mod:`pyascore_amd.ingest`."""
from pyascore_amd.ingest import (COMMON_MODS, STD_AA_MASS, IdentificationParser, MassCorrector,  # noqa: F401
                                 MokapotTXTExtractor, MzIdentMLExtractor, PepXMLExtractor, PercolatorTXTExtractor)
