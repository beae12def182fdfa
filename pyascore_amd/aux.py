"""Auxiliary scripting classes of pyAscore's ``ptm_scoring`` module, same names and methods as the
reference's Cython wrappers:

    PyBinnedSpectra                        pyascore/ptm_scoring/Spectra.pyx:8-125
    PyModifiedPeptide, PyFragmentGraph     pyascore/ptm_scoring/ModifiedPeptide.pyx:10-329
    PyLogMath, PyBinomialDist, PyPowerSetSum   pyascore/ptm_scoring/Util.pyx:6-134

They expose single steps of the algorithm -- one spectrum's window table, one peptide's fragment
walk, one binomial tail -- and are not on the GPU path (``PyAscore.score`` never calls them).  The
work is done by host C++ inside libpyascore_hip.so (include/pyascore_aux.h, csrc/aux_api.cpp).
Where the reference would abort the process (C++ ``throw`` under Cython: stepping past the last
fragment, an unknown residue or ion type, successes > trials) these raise ``ValueError`` /
``IndexError`` / ``RuntimeError`` instead.
"""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _typed(name, a, dtype, cname, allow_none=False):
    """The buffer checks Cython's ``np.ndarray[T, ndim=1, mode="c"]`` arguments make."""
    if a is None:
        if allow_none:
            return None
        raise TypeError("Argument '%s' must not be None" % name)
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


class PyBinnedSpectra:
    """Shuttles MS/MS peaks to equal sized mass bins and keeps the ``n_top`` most intense of each,
    most intense first; offers iteration over bins and ranks (Spectra.pyx:8-125).

    Attributes: ``bin_size, min_mz, max_mz, n_bins`` and, at the cursor, ``mz, intensity, n_peaks,
    bin, rank``."""

    def __init__(self, bin_size, n_top):
        self._lib = _lib.load()
        self._s = None
        if int(n_top) < 0:
            raise OverflowError("can't convert negative value to size_t")
        self._s = C.c_void_p(self._lib.pya_spectra_create(float(bin_size), int(n_top)))
        self._bin = 0
        self._rank = 0

    def __del__(self):
        if getattr(self, "_s", None):
            self._lib.pya_spectra_destroy(self._s)
            self._s = None

    def consume_spectra(self, mz_arr, int_arr):
        """Consumes the MZ and intensities of a single spectra; bin and rank go back to 0."""
        mz_arr = _typed("mz_arr", mz_arr, np.float64, "double")
        int_arr = _typed("int_arr", int_arr, np.float64, "double")
        if int_arr.size < mz_arr.size:
            raise ValueError("int_arr is shorter than mz_arr")
        rc = self._lib.pya_spectra_consume(self._s, _ptr(mz_arr), _ptr(int_arr), mz_arr.size)
        if rc:
            raise ValueError("spectrum has no peaks or no m/z window (the reference reads out of bounds here)")
        self._bin = 0
        self._rank = 0

    def _info(self):
        lo, hi, bs = C.c_float(), C.c_float(), C.c_float()
        nb, nt = C.c_uint64(), C.c_uint64()
        self._lib.pya_spectra_info(self._s, C.byref(lo), C.byref(hi), C.byref(bs), C.byref(nb), C.byref(nt))
        return lo.value, hi.value, bs.value, nb.value, nt.value

    def _peak(self):
        mz, it = C.c_double(), C.c_double()
        if self._lib.pya_spectra_peak(self._s, self._bin, self._rank, C.byref(mz), C.byref(it)):
            raise IndexError("no peak at bin %d, rank %d" % (self._bin, self._rank))   # reference: std::out_of_range
        return mz.value, it.value

    # peak access
    @property
    def mz(self):
        return self._peak()[0]

    @property
    def intensity(self):
        return self._peak()[1]

    @property
    def n_peaks(self):
        n = self._lib.pya_spectra_window_size(self._s, self._bin)
        if n < 0:
            raise IndexError("no bin %d" % self._bin)
        return int(n)

    # mutable state access: both cursors clamp at their end position (Spectra.cpp:85-105)
    @property
    def bin(self):
        return self._bin

    @bin.setter
    def bin(self, new_bin):
        self._bin = min(int(new_bin), self.n_bins)

    def reset_bin(self):
        self.bin = 0

    def next_bin(self):
        self.bin = self._bin + 1

    @property
    def rank(self):
        return self._rank

    @rank.setter
    def rank(self, new_rank):
        self._rank = min(int(new_rank), self._info()[4])

    def reset_rank(self):
        self.rank = 0

    def next_rank(self):
        self.rank = self._rank + 1

    # immutable state access
    @property
    def min_mz(self):
        return self._info()[0]

    @property
    def max_mz(self):
        return self._info()[1]

    @property
    def bin_size(self):
        return self._info()[2]

    @property
    def n_bins(self):
        return self._info()[3]


class PyModifiedPeptide:
    """Modified residues of peptides: a sequence, fixed position modifications and a number of
    unlocalized modifications that can fall on any residue of ``mod_group``.  One realization is
    encoded by a *signature*, a 0/1 vector with one entry per modifiable residue
    (ModifiedPeptide.pyx:10-157).

    Parameters: ``mod_group`` (e.g. "STY"), ``mod_mass`` (e.g. 79.966331), ``mz_error`` in Da
    (default 0.5), ``fragment_types`` (default "by")."""

    def __init__(self, mod_group, mod_mass, mz_error=.5, fragment_types="by"):
        if not isinstance(mod_group, str) or not isinstance(fragment_types, str):
            raise TypeError("mod_group and fragment_types must be str")
        self._lib = _lib.load()
        self._p = C.c_void_p(self._lib.pya_modpep_create(mod_group.encode("utf8"), float(mod_mass), float(mz_error),
                                                         fragment_types.encode("utf8")))
        self._graphs = 0

    def __del__(self):
        if getattr(self, "_p", None):
            self._lib.pya_modpep_destroy(self._p)
            self._p = None

    def _raise(self, rc):
        msg = self._lib.pya_modpep_last_error(self._p).decode("utf8", "replace")
        raise (RuntimeError if rc == _lib.PYA_ERR_STATE else ValueError)(msg or "invalid argument")

    def add_neutral_loss(self, group, mass):
        if not isinstance(group, str):
            raise TypeError("Argument 'group' has incorrect type (expected str)")
        self._lib.pya_modpep_add_neutral_loss(self._p, group.encode("utf8"), float(mass))

    def consume_peptide(self, peptide, n_of_mod, max_fragment_charge=1, aux_mod_pos=None, aux_mod_mass=None):
        """Consumes a single peptide sequence and creates its internal representation."""
        if not isinstance(peptide, str):
            raise TypeError("Argument 'peptide' has incorrect type (expected str, got %s)" % type(peptide).__name__)
        if int(n_of_mod) < 0 or int(max_fragment_charge) < 0:
            raise OverflowError("can't convert negative value to size_t")
        ap = _typed("aux_mod_pos", aux_mod_pos, np.uint32, "unsigned int", allow_none=True)
        am = _typed("aux_mod_mass", aux_mod_mass, np.float32, "float", allow_none=True)
        if ap is None or am is None:
            ap = am = None
        elif ap.size != am.size:
            raise ValueError("aux_mod_pos and aux_mod_mass differ in length")
        pep = peptide.encode("utf8")
        rc = self._lib.pya_modpep_consume_peptide(self._p, pep, len(pep), int(n_of_mod), int(max_fragment_charge),
                                                  _ptr(ap), _ptr(am), 0 if ap is None else ap.size)
        if rc:
            self._raise(rc)

    def consume_peak(self, mz, rank):
        """Adds one retained peak (m/z, rank inside its window) to the match cache
        (cpp/ModifiedPeptide.cpp:126-142; ``PyAscore.score`` feeds every retained peak this way)."""
        rc = self._lib.pya_modpep_consume_peak(self._p, float(np.float32(mz)), int(rank))
        if rc:
            raise RuntimeError("consume_peak needs a consumed peptide")

    def has_match(self, fragment_mz):
        return self.get_match(fragment_mz) is not None

    def get_match(self, fragment_mz):
        """(peak m/z, rank) of the lowest-ranked consumed peak within ``mz_error`` of a theoretical
        m/z, or None (cpp/ModifiedPeptide.cpp:144-150)."""
        mz, rank = C.c_float(), C.c_uint64()
        rc = self._lib.pya_modpep_get_match(self._p, float(np.float32(fragment_mz)), C.byref(mz), C.byref(rank))
        if rc < 0:
            raise RuntimeError("get_match needs a consumed peptide")
        return (mz.value, int(rank.value)) if rc else None

    def get_peptide(self, signature=None):
        """The modified sequence with bracketed modification masses, e.g. PEPT[80]IDEK."""
        sig = None
        if signature is not None:
            sig = _typed("signature", signature, np.uint32, "unsigned int")
            if sig.size == 0:
                sig = None
        buf = C.create_string_buffer(2048)
        n = self._lib.pya_modpep_get_peptide(self._p, _ptr(sig), 0 if sig is None else sig.size, buf, 2048)
        if n < 0:
            raise RuntimeError("get_peptide needs a consumed peptide")
        return buf.value.decode("utf8")

    def get_fragment_graph(self, fragment_type, charge_state, mode="all"):
        """A PyFragmentGraph of the given ion type ('b', 'c', 'y', 'z', 'Z') and charge."""
        if not isinstance(fragment_type, str) or not fragment_type:
            raise TypeError("fragment_type must be a non-empty str")
        return PyFragmentGraph(self, fragment_type.encode("utf8")[0], charge_state, mode)

    def get_site_determining_ions(self, sig_1, sig_2, fragment_type, max_charge):
        """The non-overlapping theoretical fragments of two assignments: a tuple of two float32
        arrays, the fragments of each that the other does not have within ``mz_error``."""
        sig_1 = _typed("sig_1", sig_1, np.uint32, "unsigned int")
        sig_2 = _typed("sig_2", sig_2, np.uint32, "unsigned int")
        n = min(sig_1.size, sig_2.size)
        t = fragment_type.encode("utf8")[:1]
        n1, n2 = C.c_uint64(), C.c_uint64()
        args = (self._p, _ptr(sig_1), _ptr(sig_2), n, t, int(max_charge))
        rc = self._lib.pya_modpep_site_ions(*args, None, 0, C.byref(n1), None, 0, C.byref(n2))
        if rc:
            raise ValueError("signatures must have one entry per modifiable residue and fragment_type one of b, c, y, z, Z")
        out = (np.zeros(n1.value, np.float32), np.zeros(n2.value, np.float32))
        self._lib.pya_modpep_site_ions(*args, _ptr(out[0]), out[0].size, C.byref(n1), _ptr(out[1]), out[1].size,
                                       C.byref(n2))
        return out


class PyFragmentGraph:
    """Traversal of the modification tree of a PyModifiedPeptide: every site assignment
    (*signature*) of one ion type and charge, and for each the theoretical fragment m/z.  b/c
    graphs walk from the N-terminus, y/z/Z graphs from the C-terminus, so the two iterate through
    the signatures in different orders (ModifiedPeptide.pyx:159-329).

    ``mode="all"`` restarts every signature at its first fragment; ``mode="reduced"`` resumes at
    the residue whose modification state changed, skipping the fragments shared with the previous
    signature."""

    def __init__(self, peptide, fragment_type, charge_state, mode="all"):
        assert mode in ("all", "reduced")
        if not isinstance(peptide, PyModifiedPeptide):
            raise TypeError("Argument 'peptide' has incorrect type (expected PyModifiedPeptide)")
        if isinstance(fragment_type, str):
            fragment_type = fragment_type.encode("utf8")[0]
        if isinstance(fragment_type, bytes):
            fragment_type = fragment_type[0]
        self.mode = mode
        self._peptide = peptide            # the C object reads the peptide's state: keep it alive
        self._lib = peptide._lib
        self._g = None
        g = self._lib.pya_fgraph_create(peptide._p, bytes([int(fragment_type)]), int(charge_state))
        if not g:
            raise ValueError("no peptide consumed yet, or unknown fragment type %r (b, c, y, z, Z)" % chr(int(fragment_type)))
        self._g = C.c_void_p(g)

    def __del__(self):
        if getattr(self, "_g", None):
            self._lib.pya_fgraph_destroy(self._g)
            self._g = None

    def _step(self, rc, what):
        if rc:
            raise RuntimeError("%s past the end (the reference aborts here)" % what)

    @property
    def fragment_type(self):
        return self._lib.pya_fgraph_type(self._g).decode("utf8")

    @property
    def charge_state(self):
        return int(self._lib.pya_fgraph_charge(self._g))

    def reset_iterator(self):
        """Resets iterator to the first position of the first signature."""
        self._lib.pya_fgraph_reset_iterator(self._g)

    def incr_signature(self):
        """Get next signature at position of last modification switch."""
        self._step(self._lib.pya_fgraph_incr_signature(self._g), "incr_signature")

    def is_signature_end(self):
        return bool(self._lib.pya_fgraph_is_signature_end(self._g))

    def reset_fragment(self):
        """Resets iterator to the first position of the current signature."""
        self._lib.pya_fgraph_reset_fragment(self._g)

    def incr_fragment(self):
        """Increment to next fragment for current signature."""
        self._step(self._lib.pya_fgraph_incr_fragment(self._g), "incr_fragment")

    def is_fragment_end(self):
        """Has the iterator reached the last fragment, i.e. the end of the peptide?"""
        return bool(self._lib.pya_fgraph_is_fragment_end(self._g))

    def is_loss(self):
        return bool(self._lib.pya_fgraph_is_loss(self._g))

    def set_signature(self, new_signature):
        """Change signature to a user specified value and reset to the first fragment."""
        sig = _typed("new_signature", new_signature, np.uint32, "unsigned int")
        if self._lib.pya_fgraph_set_signature(self._g, _ptr(sig), sig.size):
            raise ValueError("signature must have one entry per modifiable residue")

    def get_signature(self):
        """Current signature: uint64 array, one 0/1 per modifiable residue, N to C."""
        n = self._lib.pya_fgraph_get_signature(self._g, None, 0)
        out = np.zeros(max(int(n), 0), np.uint64)
        self._lib.pya_fgraph_get_signature(self._g, _ptr(out), out.size)
        return out

    def get_fragment_mz(self):
        mz = C.c_float()
        self._step(self._lib.pya_fgraph_fragment_mz(self._g, C.byref(mz)), "get_fragment_mz")
        return mz.value

    def get_fragment_size(self):
        """Size of the current fragment in number of amino acids."""
        return int(self._lib.pya_fgraph_fragment_size(self._g))

    def get_fragment_seq(self):
        """Sequence of the current fragment without modifications."""
        buf = C.create_string_buffer(256)
        self._lib.pya_fgraph_fragment_seq(self._g, buf, 256)
        return buf.value.decode("utf8")

    def iter_permutations(self):
        """Iterate through remaining signatures; yields this graph, ready for iteration."""
        while not self.is_signature_end():
            yield self
            self.incr_signature()
            if self.mode == "all":
                self.reset_fragment()

    def iter_fragments(self):
        """Iterate through remaining fragments of the current signature: (m/z, label) pairs."""
        while not self.is_fragment_end():
            label = self.fragment_type + str(self.get_fragment_size())
            result = (self.get_fragment_mz(), label)
            self.incr_fragment()
            yield result


class PyLogMath:
    """Float32 log-space helpers of the score arithmetic (Util.pyx:6-46)."""

    def __init__(self):
        self._lib = _lib.load()

    def log_sum(self, a, b):
        """log(exp(a) + exp(b)), evaluated in float32 like the scorer does."""
        return float(self._lib.pya_log_sum(float(a), float(b)))

    def log_bin_coef(self, k, n):
        """log of the binomial coefficient C(n, k)."""
        out = C.c_float()
        if int(k) < 0 or int(n) < 0:
            raise OverflowError("can't convert negative value to size_t")
        if self._lib.pya_log_bin_coef(int(k), int(n), C.byref(out)):
            raise ValueError("k must not exceed n")
        return out.value


class PyBinomialDist:
    """Binomial distribution in float32 log space (Util.pyx:48-98): ``prob`` = success probability."""

    def __init__(self, prob):
        self._lib = _lib.load()
        self._prob = float(prob)

    def _call(self, what, successes, trials):
        if int(successes) < 0 or int(trials) < 0:
            raise OverflowError("can't convert negative value to size_t")
        out = C.c_float()
        if self._lib.pya_binomial(self._prob, what, int(successes), int(trials), C.byref(out)):
            raise ValueError("successes must not exceed trials")
        return out.value

    def log_pmf(self, successes, trials):
        return self._call(0, successes, trials)

    def log_pvalue(self, successes, trials):
        """log P(X >= successes)."""
        return self._call(1, successes, trials)

    def log10_pvalue(self, successes, trials):
        return self._call(2, successes, trials)


class PyPowerSetSum:
    """Iterates, in ascending order, over 0 and the distinct sums of at most ``max_depth`` elements
    of ``target`` (Util.pyx:100-134) -- the neutral-loss combinations of a fragment."""

    def __init__(self, target=None, max_depth=0):
        self._lib = _lib.load()
        self._sums = np.zeros(1, np.float32)
        self._pos = 0
        if target is not None:
            self.reset(target, max_depth)

    def reset(self, target=None, max_depth=0):
        self._pos = 0
        if target is None:
            return
        t = _typed("target", target, np.float32, "float")
        n = self._lib.pya_power_set_sums(_ptr(t), t.size, max(int(max_depth), 0), None, 0)
        self._sums = np.zeros(int(n), np.float32)
        self._lib.pya_power_set_sums(_ptr(t), t.size, max(int(max_depth), 0), _ptr(self._sums), self._sums.size)

    def has_next(self):
        return self._pos < self._sums.size - 1

    def next(self):
        if not self.has_next():
            raise RuntimeError("next past the last sum (the reference aborts here)")
        self._pos += 1

    def get_sum(self):
        return float(self._sums[self._pos])
