"""pyascore_amd -- MI355X-native implementation of pyAscore's ``PyAscore.score`` hot path.

Public surface mirrors ``pyascore`` for this path (pyascore/__init__.py:17): ``PyAscore`` plus
the auxiliary scripting classes ``PyBinnedSpectra``, ``PyModifiedPeptide``, ``PyFragmentGraph``,
``PyLogMath``, ``PyBinomialDist``, ``PyPowerSetSum``.
"""
__version__ = "0.2.0"

_AUX = ("PyBinnedSpectra", "PyModifiedPeptide", "PyFragmentGraph", "PyLogMath", "PyBinomialDist", "PyPowerSetSum")


_INGEST = ("SpectraParser", "IdentificationParser", "MassCorrector", "COMMON_MODS", "STD_AA_MASS")


def __getattr__(name):
    if name == "PyAscore":
        from .ascore import PyAscore
        return PyAscore
    if name in _AUX:
        from . import aux
        return getattr(aux, name)
    if name in _INGEST:
        from . import ingest
        return getattr(ingest, name)
    raise AttributeError(name)
