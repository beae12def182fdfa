"""pyascore_amd/_fast.so (csrc/pyfast.c): the compiled way from PyAscore.score() into pya_score_one.  No GPU here:
the function pointer it is given is a ctypes callback with pya_score_one's signature that records what arrives and
answers with known results -- so the argument handling, the result tuple and every "not for the fast way" answer are
checked on the CPU; tests/test_gpu_parity.py runs the real thing against the checker."""
import ctypes as C

import numpy as np
import pytest

from pyascore_amd import _lib

_fast = pytest.importorskip("pyascore_amd._fast")
_fast.setup(np.ndarray)

PROTO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint64, C.POINTER(C.c_uint8),
                    C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint64, C.c_uint32,
                    C.POINTER(_lib.Results))


def _callee(seen, rc=0):
    def f(h, mz, it, n, pep, L, k, z, ap, am, n_aux, flags, out):
        seen.update(h=h, mz=[mz[i] for i in range(n)], it=[it[i] for i in range(n)], pep=bytes(pep[i] for i in range(L)),
                    k=k, z=z, aux=[(ap[i], am[i]) for i in range(n_aux)], flags=flags, max_k=out.contents.max_k)
        if rc:
            return rc
        r = out.contents
        C.cast(r.best_score, C.POINTER(C.c_float))[0] = 12.5
        C.cast(r.best_sig, C.POINTER(C.c_uint64))[0] = 0b1011
        C.cast(r.n_sig, C.POINTER(C.c_int32))[0] = 20
        for j in range(r.max_k):
            C.cast(r.ascores, C.POINTER(C.c_float))[j] = 1.5 + j
            C.cast(r.alt_mask, C.POINTER(C.c_uint64))[j] = 1 << (3 * j)
        return 0
    cb = PROTO(f)
    return cb, C.cast(cb, C.c_void_p).value


def test_arguments_arrive_and_results_come_back():
    seen = {}
    cb, addr = _callee(seen)
    mz = np.array([100.5, 200.25, 300.125])
    it = np.array([1.0, 2.0, 3.0])
    r = _fast.score_one(addr, 0x1234, mz, it, "PEPSTIDE", 2, np.int64(3), None, None)
    assert seen == dict(h=0x1234, mz=list(mz), it=list(it), pep=b"PEPSTIDE", k=2, z=3, aux=[], flags=0, max_k=2)
    assert r[:4] == (0, 12.5, 0b1011, 20) and r[6] == 2
    assert np.frombuffer(r[4], np.float32).tolist() == [1.5, 2.5]
    assert np.frombuffer(r[5], np.uint64).tolist() == [1, 8]
    # fixed modifications, numpy integers, n_of_mod 0 (one result column all the same)
    ap, am = np.array([1, 4], np.uint32), np.array([57.0, 15.5], np.float32)
    r = _fast.score_one(addr, 1, mz, it, "PEPSTIDE", np.int32(0), 1, ap, am)
    assert seen["aux"] == [(1, 57.0), (4, 15.5)] and seen["k"] == 0 and seen["max_k"] == 1 and r[6] == 0 and len(r[4]) == 4


def test_error_codes_come_back_alone():
    seen = {}
    cb, addr = _callee(seen, rc=_lib.PYA_ERR_STATE)
    r = _fast.score_one(addr, 1, np.zeros(2), np.zeros(2), "AS", 1, 1, None, None)
    assert r == (_lib.PYA_ERR_STATE,)


@pytest.mark.parametrize("case", ["list", "f32", "2d", "strided", "lengths", "bytes peptide", "negative", "too many", "float n",
                                  "aux dtype", "aux lengths", "no handle", "readonly ok"])
def test_what_the_fast_way_does_not_take(case):
    seen = {}
    cb, addr = _callee(seen)
    mz, it, pep, k, z, ap, am, h = np.zeros(4), np.zeros(4), "ASTK", 1, 1, None, None, 7
    if case == "list": mz = [0.0] * 4
    if case == "f32": it = np.zeros(4, np.float32)
    if case == "2d": mz = np.zeros((2, 2))
    if case == "strided": mz = np.zeros(8)[::2]
    if case == "lengths": it = np.zeros(5)
    if case == "bytes peptide": pep = b"ASTK"
    if case == "negative": k = -1
    if case == "too many": k = 65
    if case == "float n": z = 1.0
    if case == "aux dtype": ap, am = np.zeros(1, np.int64), np.zeros(1, np.float32)
    if case == "aux lengths": ap, am = np.zeros(2, np.uint32), np.zeros(1, np.float32)
    if case == "no handle": h = 0
    if case == "readonly ok":
        mz.setflags(write=False)
        assert _fast.score_one(addr, h, mz, it, pep, k, z, ap, am)[0] == 0 and seen["k"] == 1
        return
    assert _fast.score_one(addr, h, mz, it, pep, k, z, ap, am) is None and not seen
