"""TEST INFRASTRUCTURE: carries the parity suite's route / debug switches to a scorer.

The library takes no route from the environment (include/pyascore_debug.h).  The tests and the probes under scripts/
still NAME a route by the PYA_* variables they set (monkeypatch.setenv, os.environ): `from_env(scorer)` hands the
current values of exactly these names to the scorer through PyAscore.set_debug, and `install()` makes every scorer
created afterwards do that on creation.  tests/conftest.py installs it for the test session; nothing under
pyascore_amd/ imports this module."""
import os

SWITCHES = ("PYA_NO_PLAIN", "PYA_NO_FUSED", "PYA_NO_BIG", "PYA_NO_TINY", "PYA_NO_PREFIX", "PYA_NO_CHUNKS",
            "PYA_NO_UPLOAD_THREAD", "PYA_ONE_PEAK_CLASS", "PYA_PEAK_CLASSES", "PYA_ONE_LDS_CLASS", "PYA_SORT_ROOM",
            "PYA_NO_BIG_INLINE", "PYA_NO_LOC_HASH", "PYA_NO_NODES", "PYA_DEBUG", "PYA_PLAIN_MIN", "PYA_BIG_MIN_N",
            "PYA_TINY_MAX", "PYA_SORT_ROOM_MAX", "PYA_SB", "PYA_GTP", "PYA_HASH_PP", "PYA_NODE_CAP", "PYA_NO_CNT",
            "PYA_SLOW_NULL_STREAM", "PYA_BIN_SELECT_MIN", "PYA_BIN_SELECT_SCAP", "PYA_NO_FORK")
# (PYA_WORKSPACE_MB, PYA_CHUNK_MB, PYA_HOST_TIMING, PYA_STAMPS are the four variables the library reads itself)


def from_env(scorer):
    """Re-reads the library's own four variables, then sets every switch named in the environment (and only those:
    the rest are back at their production defaults)."""
    scorer.reload_env()
    for name in SWITCHES:
        v = os.environ.get(name)
        if v is not None:
            scorer.set_debug(name, v)
    return scorer


_installed = False


def install():
    global _installed
    if _installed:
        return
    from pyascore_amd import ascore
    plain_init = ascore.PyAscore.__init__

    def init_with_switches(self, *a, **kw):
        plain_init(self, *a, **kw)
        if any(os.environ.get(n) is not None for n in SWITCHES):
            from_env(self)

    ascore.PyAscore.__init__ = init_with_switches
    _installed = True
