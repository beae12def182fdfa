"""Device-resident scoring: spectra stay in HBM, results land in HBM, caller owns the stream.

Thin Python front for the pya_plan_* entry points of include/pyascore_hip.h.  PyTorch is used
only as plumbing (device allocations, the current HIP stream, torch.distributed); nothing here
computes with torch.
"""
import ctypes as C

import numpy as np

from . import _lib
from .ascore import PyAscore, _as_ptr


class DevicePlan:
    """One batch planned once (host pre-pass, tables, workspace), runnable many times.

    ``spectra`` are two float64 CUDA/HIP tensors (m/z, intensity) laid out as the batch's
    ``peak_off`` says.  ``run()`` enqueues the three kernels on torch's current stream and
    returns the result tensors (device).  ``max_k`` widens the per-site result rows beyond this
    batch's own largest n_of_mod: ranks that gather fixed-size records pass the job-wide value."""

    def __init__(self, scorer, batch, timing=False, max_k=None):
        import torch
        if not isinstance(scorer, PyAscore):
            raise TypeError("scorer must be a pyascore_amd.PyAscore")
        self._torch = torch
        self.scorer = scorer
        self._lib = scorer._lib
        self.n_psm = int(batch["n_psm"])
        self.max_k = max(1, int(np.max(batch["n_of_mod"]))) if self.n_psm else 1
        if max_k is not None:
            if int(max_k) < self.max_k:
                raise ValueError("max_k=%d is smaller than the batch's largest n_of_mod (%d)" % (max_k, self.max_k))
            self.max_k = int(max_k)
        self.device = torch.device("cuda", scorer.device)
        self._meta = dict(
            peak_off=np.ascontiguousarray(batch["peak_off"], np.int64),
            pep=np.ascontiguousarray(batch["pep"], np.uint8),
            pep_off=np.ascontiguousarray(batch["pep_off"], np.int64),
            n_of_mod=np.ascontiguousarray(batch["n_of_mod"], np.int32),
            max_charge=np.ascontiguousarray(batch["max_charge"], np.int32),
            aux_pos=np.ascontiguousarray(batch["aux_pos"], np.uint32),
            aux_mass=np.ascontiguousarray(batch["aux_mass"], np.float32),
            aux_off=np.ascontiguousarray(batch["aux_off"], np.int64))
        m = self._meta
        b = _lib.Batch(self.n_psm, _as_ptr(m["peak_off"]), _as_ptr(m["pep"]), _as_ptr(m["pep_off"]),
                       _as_ptr(m["n_of_mod"]), _as_ptr(m["max_charge"]), _as_ptr(m["aux_pos"]),
                       _as_ptr(m["aux_mass"]), _as_ptr(m["aux_off"]))
        self._plan = C.c_void_p()
        rc = self._lib.pya_plan_create(scorer._h, C.byref(b), _lib.PYA_FLAG_TIMING if timing else 0,
                                       C.byref(self._plan))
        if rc:
            self._plan = None
            scorer._raise(rc)
        self.timing = timing
        n, k = self.n_psm, self.max_k
        with torch.cuda.device(self.device):
            self.best_score = torch.empty(n, dtype=torch.float32, device=self.device)
            self.best_sig = torch.empty(n, dtype=torch.int64, device=self.device)     # u64 bit patterns
            self.n_sig = torch.empty(n, dtype=torch.int32, device=self.device)
            self.ascores = torch.empty((n, k), dtype=torch.float32, device=self.device)
            self.alt_mask = torch.empty((n, k), dtype=torch.int64, device=self.device)
        self._res = _lib.Results(k, self.best_score.data_ptr(), self.best_sig.data_ptr(), self.n_sig.data_ptr(),
                                 self.ascores.data_ptr(), self.alt_mask.data_ptr())

    def close(self):
        if getattr(self, "_plan", None) is not None and self._plan.value:
            self._lib.pya_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    __del__ = close

    @property
    def workspace_bytes(self):
        return int(self._lib.pya_plan_workspace_bytes(self._plan))

    @property
    def total_signatures(self):
        return int(self._lib.pya_plan_total_signatures(self._plan))

    def run(self, d_mz, d_intensity):
        torch = self._torch
        for t in (d_mz, d_intensity):
            if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
                raise ValueError("spectra must be contiguous float64 device tensors")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.pya_plan_run(self._plan, d_mz.data_ptr(), d_intensity.data_ptr(), stream,
                                    C.byref(self._res))
        if rc:
            self.scorer._raise(rc)
        return self

    def timings_ms(self):
        """(bin_spectra, score_signatures, score_localize, localize) kernel-family durations of the
        last run; synchronises."""
        ms = (C.c_float * 4)()
        rc = self._lib.pya_plan_timings(self._plan, C.byref(ms))
        if rc:
            self.scorer._raise(rc)
        return tuple(float(x) for x in ms)

    def timings_sum(self):
        """((bin_spectra, score_signatures, score_localize, localize) durations in ms summed over the runs since
        the last call, number of runs).  The events sit in a ring of 128 runs; synchronises with the latest run
        only, so runs can be enqueued back to back and read afterwards."""
        ms = (C.c_double * 4)()
        n = C.c_uint32(0)
        rc = self._lib.pya_plan_timings_sum(self._plan, C.byref(ms), C.byref(n))
        if rc:
            self.scorer._raise(rc)
        return tuple(float(x) for x in ms), int(n.value)

    def check(self):
        rc = self._lib.pya_plan_check(self._plan)
        if rc:
            self.scorer._raise(rc)

    def packed_summary(self, out=None):
        """Results as one [n_psm, 4 + 3*max_k] int32 device tensor (fixed-size records for the
        gather): best_score bits, n_sig, best_sig lo/hi, then per site ascore bits, alt lo/hi.
        ``out``: a preallocated contiguous [n_psm, width] int32 tensor (e.g. a slice of the send buffer).  ONE kernel
        of the library (pya_pack_records) on torch's current stream writes the records in place: no framework kernel
        and no intermediate tensor inside the step (r05: torch.cat)."""
        torch = self._torch
        width = 4 + 3 * self.max_k
        if out is None:
            out = torch.empty((self.n_psm, width), dtype=torch.int32, device=self.device)
        if out.dtype != torch.int32 or tuple(out.shape) != (self.n_psm, width) or not out.is_contiguous() or not out.is_cuda:
            raise ValueError("out must be a contiguous int32 device tensor of shape (%d, %d)" % (self.n_psm, width))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.pya_pack_records(self.scorer._h, C.byref(self._res), self.n_psm, self.max_k, out.data_ptr(), stream)
        if rc:
            self.scorer._raise(rc)
        return out


def unpack_summary(packed, max_k):
    """Inverse of DevicePlan.packed_summary on a host int32 array -> dict of numpy arrays."""
    p = np.ascontiguousarray(packed, dtype=np.int32)
    k = max_k
    return dict(
        best_score=p[:, 0].copy().view(np.float32),
        n_sig=p[:, 1].copy(),
        best_sig=np.ascontiguousarray(p[:, 2:4]).view(np.uint64).reshape(-1),
        ascores=np.ascontiguousarray(p[:, 4:4 + k]).view(np.float32),
        alt_mask=np.ascontiguousarray(p[:, 4 + k:4 + 3 * k]).view(np.uint64).reshape(-1, k),
    )
