"""``PyAscore`` -- the reference's Python surface for the PTM-localisation scorer
(pyascore/ptm_scoring/Ascore.pyx:12-288), backed by the MI355X kernels through the C ABI of
include/pyascore_hip.h.  Same constructor, ``add_neutral_loss``, ``score``, properties and
``calculate_ambiguity``; plus ``score_batch`` (``score`` is a batch of one).

Deviations, all towards *more* defined behaviour: inputs the reference
would abort or read out of bounds on (unknown residue, empty spectrum, ``n_top < 10``,
``max_fragment_charge < 1``, mismatched array lengths) raise ``ValueError`` here.
"""
import ctypes as C
import os

import numpy as np

from . import _lib


def _as_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check_f64(name, a):
    if a is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % a.dtype)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return a


def _check_typed(name, a, dtype, cname):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % a.ndim)
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'" % (cname, a.dtype))
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return a


try:                                  # the compiled way into pya_score_one (csrc/pyfast.c); ctypes without it
    from . import _fast
    _fast.setup(np.ndarray)
except ImportError:                   # not built for this interpreter
    _fast = None

_NO_U32 = np.zeros(0, np.uint32)
_NO_F32 = np.zeros(0, np.float32)


class _Last(dict):
    """What the last score() call left: the scalars as they came back, the arrays made from the raw bytes the first
    time something reads them (a loop that only reads best_score never builds them)."""

    def __missing__(self, key):
        if key == "pep":
            v = np.frombuffer(self["peptide"].encode("utf8"), dtype=np.uint8)
        elif key == "ascores":
            v = np.frombuffer(self["_asc"], dtype=np.float32)
        elif key == "alt_mask":
            v = np.frombuffer(self["_alt"], dtype=np.uint64)
        else:
            raise KeyError(key)
        self[key] = v
        return v


class PyAscore:
    """Scores the localization of post translational modifications (Ascore.pyx:12-59).

    Parameters
    ----------
    bin_size : float
        Size in MZ of each bin
    n_top : int
        Number of top peaks to retain in each bin: 10 (what everything is built for and the reference's command
        line passes) to 16 (every PSM then goes through the general kernel: same results, far slower)
    mod_group : str
        Residues that can carry the unlocalized modification, e.g. "STY" ('n'/'c' = termini)
    mod_mass : float
        Mass of the unlocalized modification, e.g. 79.966331
    mz_error : float
        Matching tolerance in Da (default 0.5)
    fragment_types : str
        Ion types to score, subset of b, c, y, z, Z (default "by")
    device : int, optional (keyword only)
        HIP device ordinal; defaults to LOCAL_RANK or 0.
    """

    def __init__(self, bin_size, n_top, mod_group, mod_mass, mz_error=.5, fragment_types="by", *,
                 device=None):
        if not isinstance(mod_group, str) or not isinstance(fragment_types, str):
            raise TypeError("mod_group and fragment_types must be str")
        self._lib = _lib.load()
        self._h = None
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self._mod_group = mod_group
        self._cfg_strings = (mod_group.encode("utf8"), fragment_types.encode("utf8"))
        cfg = _lib.Config(float(bin_size), int(n_top), self._cfg_strings[0], float(mod_mass),
                          float(mz_error), self._cfg_strings[1], int(device))
        h = C.c_void_p()
        rc = self._lib.pya_create(C.byref(cfg), C.byref(h))
        self._h = h if h.value else None
        if rc:
            self._raise(rc)
        self._h_addr = int(h.value)
        self._score_one_addr = C.cast(self._lib.pya_score_one, C.c_void_p).value
        self.device = int(device)
        self._n_top = int(n_top)
        self._last = None            # summary of the last score() call
        self._batch_n = None         # PSMs of the batch retained by score_batch(keep=True)
        self._budget = 0             # set_workspace_budget (0 = the library's default, 6 GiB)
        self._lazy_batch = None      # a retained batch too big for the device: re-scored range by range on demand
        self._one = None             # preallocated batch-of-one scaffolding of score()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and getattr(self, "_lib", None) is not None:
            self._lib.pya_destroy(h)
            self._h = None
            self._h_addr = 0

    def _raise(self, rc):
        msg = self._lib.pya_last_error(self._h).decode("utf8", "replace") if self._h else "pya_create failed"
        if rc in (_lib.PYA_ERR_ARG, _lib.PYA_ERR_PSM, _lib.PYA_ERR_LIMIT):
            raise ValueError(msg)
        raise RuntimeError(msg)

    # ------------------------------------------------------------------------------------------
    def add_neutral_loss(self, group, mass):
        """Add a neutral loss ion to any fragment containing specified amino acids
        (Ascore.pyx:81-99).  Upper case = unmodified residue, lower case = modified residue."""
        if not isinstance(group, str):
            raise TypeError("Argument 'group' has incorrect type (expected str)")
        self._ensure_kept()      # (the last score() PSM's records, if they are still to be produced: under the old settings)
        rc = self._lib.pya_add_neutral_loss(self._h, group.encode("utf8"), float(mass))
        if rc:
            self._raise(rc)

    def reload_env(self):
        """Re-reads the four environment variables the library knows (PYA_WORKSPACE_MB, PYA_CHUNK_MB, PYA_HOST_TIMING,
        PYA_STAMPS: sizes and diagnostics, read once when the scorer is created) and puts every debug switch back to
        its production default."""
        self._ensure_kept()
        rc = self._lib.pya_reload_env(self._h)
        if rc:
            self._raise(rc)

    def set_debug(self, key, value=None):
        """TEST-ONLY (include/pyascore_debug.h): one debug switch of this scorer -- force a kernel route, make a kernel
        decline its work, resize a table.  ``value=None`` restores the production default.  Nothing in the
        environment selects a route; the parity suite sets the switches through this call."""
        self._ensure_kept()
        rc = self._lib.pya_set_debug(self._h, key.encode("ascii"), None if value is None else str(value).encode("ascii"))
        if rc:
            self._raise(rc)

    def set_workspace_budget(self, n_bytes):
        """Device memory one ``score_batch`` call may hold at a time (default 6 GiB; 0 restores it).
        Bigger calls are cut into chunks of consecutive PSMs and pipelined (upload of the next chunk
        under the kernels of the current one); results do not depend on the cut."""
        rc = self._lib.pya_set_workspace_budget(self._h, int(n_bytes))
        if rc:
            self._raise(rc)
        self._budget = int(n_bytes)

    def _retained_bytes(self, arrs):
        """Device bytes a retained (keep=True) plan of this batch holds, per PSM: 16 per raw peak (spectra) + 8 per
        peak (retained table) + per site assignment the score, the order and the count record (4 + 4 + 4 x its words:
        32 bytes for n_top = 10, 44 for 16; with n_top > 10 also the general kernel's sort area, 8 more) + grid,
        descriptor, results."""
        from .shard import comb_table, count_sites
        n_sites = np.clip(count_sites(arrs, self._mod_group), 0, 64)
        k = arrs["n_of_mod"].astype(np.int64)
        sigs = np.where((k >= 0) & (k <= n_sites), comb_table()[n_sites, np.clip(k, 0, 64)], 0.0)
        peaks = np.diff(arrs["peak_off"]).astype(np.float64)
        rec_words = (self._n_top + 1) // 2 + 1
        per_sig = 8.0 + 4.0 * rec_words + (8.5 if self._n_top != 10 else 0.0)
        return 24.0 * peaks + per_sig * sigs + 1024.0

    def score(self, mz_arr, int_arr, peptide, n_of_mod, max_fragment_charge=1, aux_mod_pos=None,
              aux_mod_mass=None):
        """Consume spectra and associated peptide information and score PTM localization
        (Ascore.pyx:103-152)."""
        if _fast is not None:
            # the compiled way: arguments as they are; None = something it does not take (types, layouts, lengths,
            # negative numbers, more than 64 modifications): the checked way below raises what the reference raises
            r = _fast.score_one(self._score_one_addr, self._h_addr, mz_arr, int_arr, peptide, n_of_mod, max_fragment_charge,
                                aux_mod_pos, aux_mod_mass)
            if r is not None and r[0] == 0:
                have_aux = aux_mod_pos is not None and aux_mod_mass is not None
                self._batch_n = None
                self._last = _Last(peptide=peptide, k=r[6], aux_pos=aux_mod_pos.copy() if have_aux else _NO_U32,
                                   aux_mass=aux_mod_mass.copy() if have_aux else _NO_F32, best_score=r[1], best_sig=r[2],
                                   n_sig=r[3], _asc=r[4], _alt=r[5], lazy=True)
                return
            # (an error code, or PYA_ERR_STATE = "not for the one-PSM kernel": the way below handles both)
        mz_arr = _check_f64("mz_arr", mz_arr)
        int_arr = _check_f64("int_arr", int_arr)
        if not isinstance(peptide, str):
            raise TypeError("Argument 'peptide' has incorrect type (expected str, got %s)"
                            % type(peptide).__name__)
        if int_arr.size != mz_arr.size:
            raise ValueError("mz_arr and int_arr differ in length (%d vs %d)" % (mz_arr.size, int_arr.size))
        if int(n_of_mod) < 0 or int(max_fragment_charge) < 0:
            raise OverflowError("can't convert negative value to size_t")
        pep = np.frombuffer(peptide.encode("utf8"), dtype=np.uint8)
        if aux_mod_pos is not None and aux_mod_mass is not None:
            ap = _check_typed("aux_mod_pos", aux_mod_pos, np.uint32, "unsigned int")
            am = _check_typed("aux_mod_mass", aux_mod_mass, np.float32, "float")
            if ap.size != am.size:
                raise ValueError("aux_mod_pos and aux_mod_mass differ in length")
        else:
            ap = np.zeros(0, np.uint32)
            am = np.zeros(0, np.float32)
        # One PSM through pya_score_one (no plan, no copies: pinned spectrum block, scalars in the kernel
        # arguments, results polled from pinned memory).  The per-signature records behind ``pep_scores`` and
        # ``calculate_ambiguity`` are retained on demand (_ensure_kept): a loop over PSMs that reads only
        # best_sequence / best_score / ascores / alt_sites, like the reference's command line, never pays for them.
        one = self._one
        k = max(1, int(n_of_mod))
        if one is None or one["k"] < k:
            one = dict(k=k, peak_off=np.zeros(2, np.int64), pep_off=np.zeros(2, np.int64),
                       aux_off=np.zeros(2, np.int64), n_of_mod=np.zeros(1, np.int32),
                       max_charge=np.zeros(1, np.int32), best_score=np.zeros(1, np.float32),
                       best_sig=np.zeros(1, np.uint64), n_sig=np.zeros(1, np.int32),
                       ascores=np.zeros((1, k), np.float32), alt_mask=np.zeros((1, k), np.uint64))
            one["batch"] = _lib.Batch(1, _as_ptr(one["peak_off"]), None, _as_ptr(one["pep_off"]),
                                      _as_ptr(one["n_of_mod"]), _as_ptr(one["max_charge"]), None, None,
                                      _as_ptr(one["aux_off"]))
            one["results"] = _lib.Results(k, _as_ptr(one["best_score"]), _as_ptr(one["best_sig"]),
                                          _as_ptr(one["n_sig"]), _as_ptr(one["ascores"]), _as_ptr(one["alt_mask"]))
            self._one = one
        one["ascores"][:] = 0
        one["alt_mask"][:] = 0
        lazy = True
        rc = _lib.PYA_ERR_STATE
        if one["k"] <= 64:
            rc = self._lib.pya_score_one(self._h, mz_arr.ctypes.data, int_arr.ctypes.data, mz_arr.size, pep.ctypes.data,
                                         pep.size, int(n_of_mod), int(max_fragment_charge), ap.ctypes.data, am.ctypes.data,
                                         ap.size, 0, C.byref(one["results"]))
        if rc == _lib.PYA_ERR_STATE and not self._lib.pya_last_error(self._h):
            # (more than 8 fixed modifications: a retained batch of one through the plan machinery)
            lazy = False
            one["peak_off"][1] = mz_arr.size
            one["pep_off"][1] = pep.size
            one["aux_off"][1] = ap.size
            one["n_of_mod"][0] = n_of_mod
            one["max_charge"][0] = max_fragment_charge
            b = one["batch"]
            b.pep = pep.ctypes.data
            b.aux_pos = ap.ctypes.data
            b.aux_mass = am.ctypes.data
            rc = self._lib.pya_score_batch(self._h, C.byref(b), mz_arr.ctypes.data, int_arr.ctypes.data,
                                           _lib.PYA_FLAG_KEEP, C.byref(one["results"]))
        if rc:
            self._last = None
            self._batch_n = None
            self._raise(rc)
        self._batch_n = None if lazy else 1
        self._last = dict(pep=pep, peptide=peptide, k=int(n_of_mod), aux_pos=ap.copy(), aux_mass=am.copy(),
                          best_score=float(one["best_score"][0]), best_sig=int(one["best_sig"][0]),
                          n_sig=int(one["n_sig"][0]), ascores=one["ascores"][0].copy(),
                          alt_mask=one["alt_mask"][0].copy(), lazy=lazy)

    def _ensure_kept(self):
        """Retains the per-signature records of the last ``score()`` PSM (its inputs are still where
        pya_score_one staged them)."""
        last = self._last
        if last is not None and last.get("lazy"):
            rc = self._lib.pya_rescore_last_keep(self._h)
            if rc == _lib.PYA_ERR_STATE and not self._lib.pya_last_error(self._h):
                raise RuntimeError("the records of this PSM do not fit the one-PSM kernel; score it with "
                                   "score_batch(keep=True) to read pep_scores")
            if rc:
                self._raise(rc)
            last["lazy"] = False
            self._batch_n = 1

    def score_batch(self, batch, keep=False, skip_invalid=False):
        """Scores a CSR batch (see pyascore_amd.synth) in one call.

        Returns dict(best_score f32[n], best_sig u64[n], n_sig i32[n], ascores f32[n, max_k],
        alt_mask u64[n, max_k]); row i holds what the reference's properties would hold after
        ``score()`` of PSM i (ascores beyond n_of_mod[i] are 0).

        ``skip_invalid=True``: a PSM that is invalid (unknown residue, empty spectrum, ...) or beyond
        a documented limit of this implementation does not fail the call; it gets best_score -1,
        n_sig -1 and a non-zero code in the extra ``status`` array (include/pyascore_hip.h PYA_PSM_*),
        and ``status_message`` describes the first such PSM."""
        # the records behind pep_scores / calculate_ambiguity of the last score() PSM are produced on demand by
        # replaying what score() staged in the library: before another call reuses that staging, produce them
        self._ensure_kept()
        n = int(batch["n_psm"])
        mz = np.ascontiguousarray(batch["mz"], np.float64)
        it = np.ascontiguousarray(batch["intensity"], np.float64)
        arrs = dict(
            peak_off=np.ascontiguousarray(batch["peak_off"], np.int64),
            pep=np.ascontiguousarray(batch["pep"], np.uint8),
            pep_off=np.ascontiguousarray(batch["pep_off"], np.int64),
            n_of_mod=np.ascontiguousarray(batch["n_of_mod"], np.int32),
            max_charge=np.ascontiguousarray(batch["max_charge"], np.int32),
            aux_pos=np.ascontiguousarray(batch["aux_pos"], np.uint32),
            aux_mass=np.ascontiguousarray(batch["aux_mass"], np.float32),
            aux_off=np.ascontiguousarray(batch["aux_off"], np.int64))
        if arrs["peak_off"].size != n + 1 or arrs["pep_off"].size != n + 1 or arrs["aux_off"].size != n + 1:
            raise ValueError("offset arrays must have n_psm + 1 entries")
        if n and (mz.size < arrs["peak_off"][-1] or it.size < arrs["peak_off"][-1]):
            raise ValueError("peak_off runs past the end of the spectrum arrays")
        max_k = max(1, int(arrs["n_of_mod"].max())) if n else 1
        out = dict(best_score=np.zeros(n, np.float32), best_sig=np.zeros(n, np.uint64),
                   n_sig=np.zeros(n, np.int32), ascores=np.zeros((n, max_k), np.float32),
                   alt_mask=np.zeros((n, max_k), np.uint64))
        if n == 0:
            return out
        b = _lib.Batch(n, _as_ptr(arrs["peak_off"]), _as_ptr(arrs["pep"]), _as_ptr(arrs["pep_off"]),
                       _as_ptr(arrs["n_of_mod"]), _as_ptr(arrs["max_charge"]), _as_ptr(arrs["aux_pos"]),
                       _as_ptr(arrs["aux_mass"]), _as_ptr(arrs["aux_off"]))
        r = _lib.Results(max_k, _as_ptr(out["best_score"]), _as_ptr(out["best_sig"]), _as_ptr(out["n_sig"]),
                         _as_ptr(out["ascores"]), _as_ptr(out["alt_mask"]))
        # A retained batch is ONE plan on the device.  When its records do not fit the workspace budget the batch is
        # scored without them (chunked and pipelined like any big call) and batch_pep_scores() re-scores the range it
        # is asked for, a budget's worth of PSMs at a time: the export works for any batch size.
        self._lazy_batch = None
        lazy_keep = False
        if keep and n > 1:
            budget = int(self._lib.pya_get_workspace_budget(self._h))
            try:
                per_psm = self._retained_bytes(dict(arrs, n_of_mod=arrs["n_of_mod"]))
                lazy_keep = float(per_psm.sum()) > 0.8 * budget
            except (IndexError, ValueError):
                lazy_keep = False            # malformed offsets: the library's own validation reports them
        flags = (_lib.PYA_FLAG_KEEP if keep and not lazy_keep else 0) | (_lib.PYA_FLAG_SKIP_INVALID if skip_invalid else 0)
        rc = self._lib.pya_score_batch(self._h, C.byref(b), _as_ptr(mz), _as_ptr(it), flags, C.byref(r))
        if rc:
            self._raise(rc)
        self._batch_n = n if keep else None
        if keep:
            self._last = None        # the handle's retained plan now belongs to this batch, not to score()'s PSM
        if lazy_keep:
            # (references, not copies: a batch this size is gigabytes.  The arrays must not be modified before
            # batch_pep_scores() has been read -- it re-scores the ranges it is asked for from them.)
            self._lazy_batch = dict(arrs, mz=mz, intensity=it, n_psm=n, per_psm=per_psm, budget=budget)
        if skip_invalid:
            out["status"] = np.zeros(n, np.int32)
            rc = self._lib.pya_last_batch_status(self._h, _as_ptr(out["status"]), n)
            if rc:
                self._raise(rc)
            out["status_message"] = (self._lib.pya_last_error(self._h).decode("utf8", "replace")
                                     if out["status"].any() else "")
        return out

    def format_batch(self, batch, sig_bits, valid=None, rec_psm=None):
        """Modified-sequence strings (``best_sequence`` / the ``sequence`` of ``pep_scores`` records) for
        many localisations in ONE library call: record r is the localisation ``sig_bits[r]`` of PSM
        ``rec_psm[r]`` of ``batch`` (default: record r belongs to PSM r).  Records with ``valid[r] <= 0``
        (pass ``n_sig``) come back as ''.  Returns a list of str."""
        sig_bits = np.ascontiguousarray(sig_bits, np.uint64)
        n_rec = sig_bits.size
        rp = None if rec_psm is None else np.ascontiguousarray(rec_psm, np.int64)
        va = None if valid is None else np.ascontiguousarray(valid, np.int32)
        if (rp is not None and rp.size != n_rec) or (va is not None and va.size != n_rec):
            raise ValueError("rec_psm / valid must have one entry per record")
        arrs = [np.ascontiguousarray(batch["peak_off"], np.int64), np.ascontiguousarray(batch["pep"], np.uint8),
                np.ascontiguousarray(batch["pep_off"], np.int64), np.ascontiguousarray(batch["n_of_mod"], np.int32),
                np.ascontiguousarray(batch["max_charge"], np.int32), np.ascontiguousarray(batch["aux_pos"], np.uint32),
                np.ascontiguousarray(batch["aux_mass"], np.float32), np.ascontiguousarray(batch["aux_off"], np.int64)]
        if rp is None and n_rec != int(batch["n_psm"]):
            raise ValueError("one localisation per PSM expected")
        b = _lib.Batch(int(batch["n_psm"]), *[_as_ptr(a) for a in arrs])
        off = np.zeros(n_rec + 1, np.int64)
        rc = self._lib.pya_format_peptides(self._h, C.byref(b), n_rec, _as_ptr(rp), _as_ptr(sig_bits), _as_ptr(va),
                                           _as_ptr(off), None, 0)
        if rc:
            raise ValueError("record refers to a PSM outside the batch")
        buf = np.zeros(max(int(off[-1]), 1), np.uint8)
        rc = self._lib.pya_format_peptides(self._h, C.byref(b), n_rec, _as_ptr(rp), _as_ptr(sig_bits), _as_ptr(va),
                                           _as_ptr(off), _as_ptr(buf), buf.size)
        if rc:
            self._raise(rc)
        text = buf.tobytes().decode("utf8")
        return [text[off[r]:off[r + 1]] for r in range(n_rec)]

    # ------------------------------------------------------------------------------------------
    def _format(self, last, bits, sig_len):
        buf = C.create_string_buffer(1024)
        n = self._lib.pya_format_peptide(self._h, _as_ptr(last["pep"]), last["pep"].size, last["k"],
                                         _as_ptr(last["aux_pos"]), _as_ptr(last["aux_mass"]),
                                         last["aux_pos"].size, int(bits), int(sig_len), buf, 1024)
        if n < 0:
            self._raise(n)
        return buf.value.decode("utf8")

    def _n_sites(self, last):
        ns = C.c_int32()
        self._lib.pya_count_sites(self._h, _as_ptr(last["pep"]), last["pep"].size, C.byref(ns), None)
        return ns.value

    @property
    def best_sequence(self):
        last = self._last
        if last is None or last["n_sig"] <= 0:
            return ""
        return self._format(last, last["best_sig"], self._n_sites(last))

    @property
    def best_score(self):
        return -1.0 if self._last is None else self._last["best_score"]

    @property
    def pep_scores(self):
        last = self._last
        if last is None or last["n_sig"] <= 0:
            return []
        self._ensure_kept()
        n = last["n_sig"]
        ns = self._n_sites(last)
        bits = np.zeros(n, np.uint64)
        counts = np.zeros((n, self._n_top), np.int32)
        scores = np.zeros((n, self._n_top), np.float32)
        ws = np.zeros(n, np.float32)
        nfrag = np.zeros(n, np.int32)
        got = C.c_uint64()
        rc = self._lib.pya_get_pep_scores(self._h, 0, n, C.byref(got), _as_ptr(bits), _as_ptr(counts),
                                          _as_ptr(scores), _as_ptr(ws), _as_ptr(nfrag))
        if rc:
            self._raise(rc)
        out = []
        for i in range(int(got.value)):
            b = int(bits[i])
            out.append(dict(signature=np.array([(b >> j) & 1 for j in range(ns)], dtype=np.int32),
                            counts=counts[i].copy(), scores=scores[i].copy(),
                            weighted_score=float(ws[i]), total_fragments=int(nfrag[i]),
                            sequence=self._format(last, b, ns)))
        return out

    def batch_pep_scores(self, begin=0, end=None, batch=None):
        """All localisations of PSMs [begin, end) of the last ``score_batch(..., keep=True)``, in the
        reference's sorted order, as CSR arrays (bulk form of ``pep_scores``, Ascore.pyx:241-252):
        dict(rec_off i64[n+1], sig_bits u64[R] (bit j = j-th modifiable residue), counts i32[R, n_top],
        scores f32[R, n_top], weighted_score f32[R], total_fragments i32[R]); records of PSM i are
        rows rec_off[i - begin] : rec_off[i - begin + 1].  With ``batch`` (the scored batch) the
        records' ``sequence`` strings are added, all formatted in one library call."""
        if self._batch_n is None:
            raise RuntimeError("no batch retained: call score_batch(batch, keep=True) first")
        end = self._batch_n if end is None else int(end)
        begin = int(begin)
        if not 0 <= begin <= end <= self._batch_n:
            raise ValueError("PSM range outside the retained batch")
        if self._lazy_batch is not None:
            out = self._lazy_pep_scores(begin, end)
        else:
            out = self._range_pep_scores(begin, end)
        if batch is not None:
            rec_psm = np.repeat(np.arange(begin, end, dtype=np.int64), np.diff(out["rec_off"]))
            out["sequence"] = self.format_batch(batch, out["sig_bits"], rec_psm=rec_psm)
        return out

    def _range_pep_scores(self, begin, end):
        off = np.zeros(end - begin + 1, np.int64)
        rc = self._lib.pya_get_pep_scores_range(self._h, begin, end, 0, _as_ptr(off), None, None, None, None, None)
        if rc:
            self._raise(rc)
        total = int(off[-1])
        out = dict(rec_off=off, sig_bits=np.zeros(total, np.uint64), counts=np.zeros((total, self._n_top), np.int32),
                   scores=np.zeros((total, self._n_top), np.float32), weighted_score=np.zeros(total, np.float32),
                   total_fragments=np.zeros(total, np.int32))
        if total:
            rc = self._lib.pya_get_pep_scores_range(self._h, begin, end, total, _as_ptr(off),
                                                    _as_ptr(out["sig_bits"]), _as_ptr(out["counts"]),
                                                    _as_ptr(out["scores"]), _as_ptr(out["weighted_score"]),
                                                    _as_ptr(out["total_fragments"]))
            if rc:
                self._raise(rc)
        return out

    def _lazy_pep_scores(self, begin, end):
        """Records of PSMs [begin, end) of a retained batch that was too big to keep on the device: the range is
        re-scored in pieces that fit the budget, each retained, exported and dropped."""
        from .synth import slice_batch
        lb = self._lazy_batch
        cost = np.concatenate([[0.0], np.cumsum(lb["per_psm"])])
        parts, lo = [], begin
        while lo < end:
            hi = int(np.searchsorted(cost, cost[lo] + 0.5 * lb["budget"], side="right")) - 1
            hi = min(max(hi, lo + 1), end)
            sub = slice_batch(lb, lo, hi)
            arrs = {k: np.ascontiguousarray(sub[k]) for k in ("peak_off", "pep", "pep_off", "n_of_mod", "max_charge",
                                                              "aux_pos", "aux_mass", "aux_off")}
            m = hi - lo
            mk = max(1, int(arrs["n_of_mod"].max()))
            tmp = dict(best_score=np.zeros(m, np.float32), best_sig=np.zeros(m, np.uint64), n_sig=np.zeros(m, np.int32),
                       ascores=np.zeros((m, mk), np.float32), alt_mask=np.zeros((m, mk), np.uint64))
            b = _lib.Batch(m, _as_ptr(arrs["peak_off"]), _as_ptr(arrs["pep"]), _as_ptr(arrs["pep_off"]),
                           _as_ptr(arrs["n_of_mod"]), _as_ptr(arrs["max_charge"]), _as_ptr(arrs["aux_pos"]),
                           _as_ptr(arrs["aux_mass"]), _as_ptr(arrs["aux_off"]))
            r = _lib.Results(mk, _as_ptr(tmp["best_score"]), _as_ptr(tmp["best_sig"]), _as_ptr(tmp["n_sig"]),
                             _as_ptr(tmp["ascores"]), _as_ptr(tmp["alt_mask"]))
            mz = np.ascontiguousarray(sub["mz"])
            it = np.ascontiguousarray(sub["intensity"])
            rc = self._lib.pya_score_batch(self._h, C.byref(b), _as_ptr(mz), _as_ptr(it),
                                           _lib.PYA_FLAG_KEEP | _lib.PYA_FLAG_SKIP_INVALID, C.byref(r))
            if rc:
                self._raise(rc)
            parts.append(self._range_pep_scores(0, m))
            lo = hi
        off = np.concatenate([[0]] + [p["rec_off"][1:] + sum(int(q["rec_off"][-1]) for q in parts[:i])
                                      for i, p in enumerate(parts)]).astype(np.int64)
        out = {k: np.concatenate([p[k] for p in parts]) for k in ("sig_bits", "counts", "scores", "weighted_score",
                                                                 "total_fragments")}
        out["rec_off"] = off
        return out

    @property
    def ascores(self):
        if self._last is None:
            return np.zeros(0, np.float32)
        return self._last["ascores"][: self._last["k"]].astype(np.float32)

    @property
    def alt_sites(self):
        if self._last is None:
            return []
        return [np.array(self.alt_positions(self._last["alt_mask"][j], self._last["pep"]), dtype=np.uint32)
                for j in range(self._last["k"])]

    def alt_positions(self, mask, pep):
        """1-based peptide positions named by one ``alt_mask`` word of a PSM with the letters ``pep`` (bytes / uint8
        array).  Bit p = residue p for peptides of up to 64 residues; for longer ones (scored by the general kernel)
        bit j = the j-th modifiable residue (include/pyascore_hip.h: pya_results.alt_mask)."""
        m = int(mask)
        pep = np.frombuffer(bytes(pep), dtype=np.uint8) if not isinstance(pep, np.ndarray) else np.ascontiguousarray(pep, np.uint8)
        if pep.size <= 64:
            return [p + 1 for p in range(64) if (m >> p) & 1]
        ns = C.c_int32(0)
        pos = np.zeros(_lib.PYA_MAX_PEPTIDE_LEN, np.uint16)
        self._lib.pya_count_sites(self._h, _as_ptr(pep), pep.size, C.byref(ns), _as_ptr(pos))
        return [int(pos[j]) + 1 for j in range(min(int(ns.value), 64)) if (m >> j) & 1]

    def calculate_ambiguity(self, ref_score, other_score):
        """Calculate ambiguity between 2 competing localizations of the last scored PSM
        (Ascore.pyx:208-230).  Inputs should come from ``pep_scores``."""
        if self._last is None:
            raise RuntimeError("calculate_ambiguity needs a scored PSM")
        self._ensure_kept()
        from_sig = lambda s: sum(1 << j for j, v in enumerate(s) if int(v))  # noqa: E731
        rs = np.ascontiguousarray(ref_score["scores"], np.float32)
        os_ = np.ascontiguousarray(other_score["scores"], np.float32)
        if rs.size != self._n_top or os_.size != self._n_top:
            raise ValueError("score containers must hold %d depth scores (n_top)" % self._n_top)
        out = C.c_float()
        rc = self._lib.pya_calculate_ambiguity(
            self._h, 0, from_sig(ref_score["signature"]), _as_ptr(rs), float(ref_score["weighted_score"]),
            from_sig(other_score["signature"]), _as_ptr(os_), float(other_score["weighted_score"]),
            C.byref(out))
        if rc:
            self._raise(rc)
        return float(out.value)
