"""The wavefront primitives every kernel leans on (csrc/device_common.hip.h), directly: prefix sums and reductions by DPP
(r05: they replaced ds_bpermute butterflies in all kernels) and the rank of a lane in a ballot, through the test-only entry
point pya_debug_wave_ops, against numpy -- random, extreme and packed values, by one full wavefront as every call site runs
them (the scan sequence runs through the lanes: a reduction over some of the lanes is not what these are)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(scorer, vals):
    vin = np.ascontiguousarray(vals, np.int32)
    out = np.zeros(263, np.int32)
    rc = scorer._lib.pya_debug_wave_ops(scorer._h, vin.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def _wrap(a):
    return (np.asarray(a, np.uint64) & np.uint64(0xffffffff)).astype(np.uint32).view(np.int32)


def test_scans_reductions_and_ranks_match_numpy():
    from pyascore_amd import PyAscore
    s = PyAscore(100.0, 10, "STY", 79.966331)
    rng = np.random.default_rng(11)
    cases = [rng.integers(0, 1000, 64), rng.integers(-2**31, 2**31 - 1, 64), np.zeros(64), np.ones(64), np.arange(64) * 65537,
             np.full(64, 2**31 - 1), np.r_[np.zeros(63), 7], np.r_[9, np.zeros(63)], rng.integers(0, 2, 64), np.arange(64)[::-1] * 2]
    cases += [rng.integers(0, 2**16, 64) * 65536 + rng.integers(0, 2**16, 64) - 2**31 for _ in range(40)]     # (packed 16-bit fields)
    for v in cases:
        v = np.asarray(v, np.int64).astype(np.int32)
        u = v.view(np.uint32).astype(np.uint64)
        out = _run(s, v)
        incl = np.cumsum(u)
        assert np.array_equal(out[:64], _wrap(incl - u)), "exclusive scan"
        assert np.array_equal(out[64:128], _wrap(np.r_[np.cumsum(u[:32]), np.cumsum(u[32:])])), "scan within the halves"
        assert np.array_equal(out[128:192], _wrap(incl)), "inclusive scan"
        odd = (v & 1).astype(np.int64)
        assert np.array_equal(out[192:256], np.cumsum(odd) - odd), "rank in a ballot"
        assert out[256] == _wrap([incl[-1]])[0] and out[257] == out[256], "total / sum"
        assert np.uint32(out[258]) == v.view(np.uint32).max() and np.uint32(out[259]) == v.view(np.uint32).min()
        first = np.flatnonzero(odd)
        assert np.uint32(out[262]) == (first[0] if first.size else 0xffffffff)
    # float maximum and minimum (finite values, infinities; the comparisons are the kernels' own `b > a ? b : a`)
    for _ in range(40):
        f = rng.normal(0, 1e3, 64).astype(np.float32)
        f[rng.integers(64)] = np.float32(-np.inf)
        f[rng.integers(64)] = np.float32(np.inf) if rng.random() < 0.5 else np.float32(0.0)
        out = _run(s, f.view(np.int32))
        got = np.array(out[260:262], np.int32).view(np.float32)
        assert got[0] == f.max() and got[1] == f.min()
