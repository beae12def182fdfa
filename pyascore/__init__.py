"""Drop-in import name: ``from pyascore import PyAscore`` (the reference's pyascore/__init__.py:17)
resolves to the MI355X-native implementation in :mod:`pyascore_amd`.

The ptm_scoring surface -- PyAscore and the auxiliary scripting classes -- and the names the
reference exports from its parsing package (pyascore/parsing/__init__.py), served by the
dependency-free readers of :mod:`pyascore_amd.ingest`."""
from pyascore_amd import __version__  # noqa: F401

_NAMES = ("PyAscore", "PyBinnedSpectra", "PyModifiedPeptide", "PyFragmentGraph", "PyLogMath",
          "PyBinomialDist", "PyPowerSetSum")
_INGEST = ("COMMON_MODS", "STD_AA_MASS", "MassCorrector", "PepXMLExtractor", "IdentificationParser",
           "MzMLExtractor", "SpectraParser")
__all__ = list(_NAMES) + list(_INGEST)


def __getattr__(name):
    if name in _NAMES:
        import pyascore_amd
        return getattr(pyascore_amd, name)
    if name in _INGEST:
        from pyascore_amd import ingest
        return getattr(ingest, name)
    if name in ("spec_parsers", "id_parsers"):
        import importlib
        return importlib.import_module("pyascore." + name)
    raise AttributeError("module 'pyascore' has no attribute %r" % name)
