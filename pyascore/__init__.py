"""Drop-in import name: ``from pyascore import PyAscore`` (the reference's pyascore/__init__.py:17)
resolves to the MI355X-native implementation in :mod:`pyascore_amd`.

Only the ptm_scoring surface is provided -- PyAscore and the auxiliary scripting classes; the
reference's file parsers (pyteomics) and CLI are out of scope."""
from pyascore_amd import __version__  # noqa: F401

_NAMES = ("PyAscore", "PyBinnedSpectra", "PyModifiedPeptide", "PyFragmentGraph", "PyLogMath",
          "PyBinomialDist", "PyPowerSetSum")
__all__ = list(_NAMES)


def __getattr__(name):
    if name in _NAMES:
        import pyascore_amd
        return getattr(pyascore_amd, name)
    raise AttributeError("module 'pyascore' has no attribute %r" % name)
