"""ctypes binding of oracle/oracle_abi.h -- TEST INFRASTRUCTURE ONLY.

Loads either implementation of the oracle ABI:
  * ``load("oracle")``  -> oracle/libascore_oracle.so  (this repo's CPU restatement)
  * ``load("ref")``     -> oracle/_ref/libascore_ref.so (reference C++ core, built here)

``OracleAscore`` mirrors the reference's ``PyAscore`` surface (Ascore.pyx:12-288) so parity
tests read like the reference's own tests.  Nothing under pyascore_amd/ imports this module.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {
    "oracle": os.path.join(_HERE, "libascore_oracle.so"),
    "ref": os.path.join(_HERE, "_ref", "libascore_ref.so"),
}
_LIBS = {}

_p = C.c_void_p
_u64 = C.c_uint64
_f = C.c_float


def available(kind):
    return os.path.exists(_PATHS[kind])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load(kind):
    if kind in _LIBS:
        return _LIBS[kind]
    lib = C.CDLL(_PATHS[kind])
    sig = {
        "orc_create": (_p, [_f, _u64, C.c_char_p, _f, _f, C.c_char_p]),
        "orc_destroy": (None, [_p]),
        "orc_add_neutral_loss": (None, [_p, C.c_char_p, _f]),
        "orc_score": (C.c_int, [_p, _p, _p, _u64, C.c_char_p, _u64, _u64, _p, _p, _u64]),
        "orc_n_pep_scores": (_u64, [_p]),
        "orc_sig_len": (_u64, [_p]),
        "orc_n_top": (_u64, [_p]),
        "orc_get_pep_scores": (None, [_p, _p, _p, _p, _p, _p]),
        "orc_best_score": (_f, [_p]),
        "orc_best_sequence": (_u64, [_p, C.c_char_p, _u64]),
        "orc_sequence": (_u64, [_p, _u64, C.c_char_p, _u64]),
        "orc_n_ascores": (_u64, [_p]),
        "orc_get_ascores": (None, [_p, _p]),
        "orc_alt_sites": (_u64, [_p, _u64, _p, _u64]),
        "orc_calculate_ambiguity": (_f, [_p, _p, _p, _f, _p, _p, _f, _u64, _u64]),
        "orc_score_batch": (C.c_int64, [_p, _u64] + [_p] * 10 + [_u64] + [_p] * 5),
        "orc_consume_spectra": (C.c_int, [_p, _p, _p, _u64]),
        "orc_binned": (_u64, [_p, _p, _p, _p, _p, _u64, _p, _p, _p]),
        "orc_consume_peptide": (C.c_int, [_p, C.c_char_p, _u64, _u64, _p, _p, _u64]),
        "orc_fragments": (_u64, [_p, C.c_char, _u64, _p, _p, _p, _p, _u64]),
        "orc_signature_order": (_u64, [_p, C.c_char, _p, _u64]),
        "orc_site_determining": (_u64, [_p, _p, _p, C.c_char, _u64, _p, _p, _p, _p, _u64]),
        "orc_get_peptide": (_u64, [_p, _p, _u64, C.c_char_p, _u64]),
        "orc_log_sum": (_f, [_f, _f]),
        "orc_log_bin_coef": (_f, [_u64, _u64]),
        "orc_binom_log_pmf": (_f, [_f, _u64, _u64]),
        "orc_binom_log_pvalue": (_f, [_f, _u64, _u64]),
        "orc_binom_log10_pvalue": (_f, [_f, _u64, _u64]),
        "orc_power_set_sums": (_u64, [_p, _u64, _u64, _p, _u64]),
        "orc_impl_name": (C.c_char_p, []),
    }
    if kind == "oracle":
        sig["orc_std_sort"] = (None, [_p, _u64, _p])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIBS[kind] = lib
    return lib


def std_sort(keys):
    """Permutation the reference's std::sort call (Ascore.cpp:141-146) applies to `keys`."""
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    perm = np.zeros(keys.size, np.uint32)
    load("oracle").orc_std_sort(_ptr(keys), keys.size, _ptr(perm))
    return perm


class OracleAscore:
    """PyAscore-shaped front over an oracle ABI library (kind = "oracle" | "ref")."""

    def __init__(self, bin_size, n_top, mod_group, mod_mass, mz_error=0.5, fragment_types="by",
                 kind="oracle"):
        self.lib = load(kind)
        self.kind = kind
        self.h = self.lib.orc_create(bin_size, n_top, mod_group.encode(), mod_mass, mz_error,
                                     fragment_types.encode())
        self._k = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_destroy(self.h)
            self.h = None

    def add_neutral_loss(self, group, mass):
        self.lib.orc_add_neutral_loss(self.h, group.encode(), mass)

    def score(self, mz_arr, int_arr, peptide, n_of_mod, max_fragment_charge=1, aux_mod_pos=None,
              aux_mod_mass=None):
        mz_arr = np.ascontiguousarray(mz_arr, dtype=np.float64)
        int_arr = np.ascontiguousarray(int_arr, dtype=np.float64)
        if aux_mod_pos is not None and aux_mod_mass is not None:
            ap = np.ascontiguousarray(aux_mod_pos, dtype=np.uint32)
            am = np.ascontiguousarray(aux_mod_mass, dtype=np.float32)
            na = ap.size
        else:
            ap = am = None
            na = 0
        rc = self.lib.orc_score(self.h, _ptr(mz_arr), _ptr(int_arr), mz_arr.size,
                                peptide.encode(), n_of_mod, max_fragment_charge, _ptr(ap),
                                _ptr(am), na)
        if rc:
            raise RuntimeError("oracle implementation threw (rc=%d)" % rc)
        self._k = n_of_mod

    # -- components -------------------------------------------------------------------------
    def consume_spectra(self, mz_arr, int_arr):
        mz_arr = np.ascontiguousarray(mz_arr, dtype=np.float64)
        int_arr = np.ascontiguousarray(int_arr, dtype=np.float64)
        rc = self.lib.orc_consume_spectra(self.h, _ptr(mz_arr), _ptr(int_arr), mz_arr.size)
        if rc:
            raise RuntimeError("consume_spectra threw")

    def binned(self, cap=65536):
        mz = np.zeros(cap)
        it = np.zeros(cap)
        b = np.zeros(cap, np.int32)
        r = np.zeros(cap, np.int32)
        lo = C.c_float()
        hi = C.c_float()
        nb = C.c_uint64()
        n = self.lib.orc_binned(self.h, _ptr(mz), _ptr(it), _ptr(b), _ptr(r), cap, C.byref(lo),
                                C.byref(hi), C.byref(nb))
        return dict(mz=mz[:n], intensity=it[:n], bin=b[:n], rank=r[:n], min_mz=lo.value,
                    max_mz=hi.value, n_bins=nb.value)

    def consume_peptide(self, peptide, n_of_mod, max_fragment_charge=1, aux_mod_pos=None,
                        aux_mod_mass=None):
        if aux_mod_pos is not None and aux_mod_mass is not None:
            ap = np.ascontiguousarray(aux_mod_pos, dtype=np.uint32)
            am = np.ascontiguousarray(aux_mod_mass, dtype=np.float32)
            na = ap.size
        else:
            ap = am = None
            na = 0
        rc = self.lib.orc_consume_peptide(self.h, peptide.encode(), n_of_mod,
                                          max_fragment_charge, _ptr(ap), _ptr(am), na)
        if rc:
            raise RuntimeError("consume_peptide threw")
        self._k = n_of_mod

    def fragments(self, ftype, charge, sig, cap=65536):
        sig = np.ascontiguousarray(sig, dtype=np.int32)
        mz = np.zeros(cap, np.float32)
        sz = np.zeros(cap, np.int32)
        ls = np.zeros(cap, np.int32)
        n = self.lib.orc_fragments(self.h, ftype.encode(), charge, _ptr(sig), _ptr(mz), _ptr(sz),
                                   _ptr(ls), cap)
        if n == 2 ** 64 - 1:
            raise RuntimeError("fragments threw")
        return mz[:n], sz[:n], ls[:n]

    def signature_order(self, ftype, cap_rows=8192):
        sl = self.lib.orc_sig_len(self.h)
        out = np.zeros((cap_rows, max(sl, 1)), np.int32)
        n = self.lib.orc_signature_order(self.h, ftype.encode(), _ptr(out), cap_rows)
        if n == 2 ** 64 - 1:
            raise RuntimeError("signature_order threw")
        return out[:n, :sl]

    def site_determining(self, sig1, sig2, ftype, max_charge, cap=65536):
        s1 = np.ascontiguousarray(sig1, dtype=np.int32)
        s2 = np.ascontiguousarray(sig2, dtype=np.int32)
        o1 = np.zeros(cap, np.float32)
        o2 = np.zeros(cap, np.float32)
        n1 = C.c_uint64()
        n2 = C.c_uint64()
        self.lib.orc_site_determining(self.h, _ptr(s1), _ptr(s2), ftype.encode(), max_charge,
                                      _ptr(o1), C.byref(n1), _ptr(o2), C.byref(n2), cap)
        return o1[:n1.value], o2[:n2.value]

    def get_peptide(self, sig=()):
        s = np.ascontiguousarray(sig, dtype=np.int32)
        buf = C.create_string_buffer(1024)
        self.lib.orc_get_peptide(self.h, _ptr(s), s.size, buf, 1024)
        return buf.value.decode()

    # -- results ----------------------------------------------------------------------------
    @property
    def best_sequence(self):
        buf = C.create_string_buffer(1024)
        self.lib.orc_best_sequence(self.h, buf, 1024)
        return buf.value.decode()

    @property
    def best_score(self):
        return float(self.lib.orc_best_score(self.h))

    def raw_pep_scores(self):
        n = self.lib.orc_n_pep_scores(self.h)
        sl = self.lib.orc_sig_len(self.h)
        nt = self.lib.orc_n_top(self.h)
        sig = np.zeros((n, sl), np.int32)
        counts = np.zeros((n, nt), np.int32)
        scores = np.zeros((n, nt), np.float32)
        ws = np.zeros(n, np.float32)
        nfrag = np.zeros(n, np.int64)
        if n:
            self.lib.orc_get_pep_scores(self.h, _ptr(sig), _ptr(counts), _ptr(scores), _ptr(ws),
                                        _ptr(nfrag))
        return dict(signature=sig, counts=counts, scores=scores, weighted_score=ws,
                    total_fragments=nfrag)

    @property
    def pep_scores(self):
        raw = self.raw_pep_scores()
        out = []
        buf = C.create_string_buffer(1024)
        for i in range(raw["weighted_score"].shape[0]):
            self.lib.orc_sequence(self.h, i, buf, 1024)
            out.append(dict(signature=raw["signature"][i].copy(), counts=raw["counts"][i].copy(),
                            scores=raw["scores"][i].copy(),
                            weighted_score=float(raw["weighted_score"][i]),
                            total_fragments=int(raw["total_fragments"][i]),
                            sequence=buf.value.decode()))
        return out

    @property
    def ascores(self):
        n = self.lib.orc_n_ascores(self.h)
        out = np.zeros(n, np.float32)
        if n:
            self.lib.orc_get_ascores(self.h, _ptr(out))
        return out

    @property
    def alt_sites(self):
        res = []
        for j in range(self._k):
            buf = np.zeros(256, np.uint32)
            n = self.lib.orc_alt_sites(self.h, j, _ptr(buf), 256)
            res.append(buf[:n].copy())
        return res

    def calculate_ambiguity(self, ref, other):
        s1 = np.ascontiguousarray(ref["signature"], dtype=np.int32)
        s2 = np.ascontiguousarray(other["signature"], dtype=np.int32)
        c1 = np.ascontiguousarray(ref["scores"], dtype=np.float32)
        c2 = np.ascontiguousarray(other["scores"], dtype=np.float32)
        return float(self.lib.orc_calculate_ambiguity(
            self.h, _ptr(s1), _ptr(c1), ref["weighted_score"], _ptr(s2), _ptr(c2),
            other["weighted_score"], s1.size, c1.size))

    def score_batch(self, batch, max_k=None):
        """batch: dict from pyascore_amd.synth.pack_batch (CSR arrays). Returns summary dict."""
        n = batch["n_psm"]
        if max_k is None:
            max_k = int(batch["n_of_mod"].max()) if n else 1
        max_k = max(max_k, 1)
        best_score = np.zeros(n, np.float32)
        best_sig = np.zeros(n, np.uint64)
        n_sig = np.zeros(n, np.int32)
        ascores = np.zeros((n, max_k), np.float32)
        alt_mask = np.zeros((n, max_k), np.uint64)
        rc = self.lib.orc_score_batch(
            self.h, n, _ptr(batch["mz"]), _ptr(batch["intensity"]), _ptr(batch["peak_off"]),
            _ptr(batch["pep"]), _ptr(batch["pep_off"]), _ptr(batch["n_of_mod"]),
            _ptr(batch["max_charge"]), _ptr(batch["aux_pos"]), _ptr(batch["aux_mass"]),
            _ptr(batch["aux_off"]), max_k, _ptr(best_score), _ptr(best_sig), _ptr(n_sig),
            _ptr(ascores), _ptr(alt_mask))
        if rc:
            raise RuntimeError("orc_score_batch failed at PSM %d" % (-rc - 1))
        return dict(best_score=best_score, best_sig=best_sig, n_sig=n_sig, ascores=ascores,
                    alt_mask=alt_mask)
