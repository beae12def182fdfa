"""ctypes binding of include/pyascore_hip.h (libpyascore_hip.so, built in-tree by build.py).

There is no fallback: if the library is missing or no HIP device is usable, importing the
scorer fails loudly.
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYA_LIB") or os.path.join(_HERE, "libpyascore_hip.so")   # PYA_LIB: A/B builds

PYA_OK, PYA_ERR_ARG, PYA_ERR_HIP, PYA_ERR_PSM, PYA_ERR_LIMIT, PYA_ERR_STATE = 0, -1, -2, -3, -4, -5
PYA_FLAG_KEEP, PYA_FLAG_TIMING, PYA_FLAG_SKIP_INVALID = 1, 2, 4
PYA_MAX_PEPTIDE_LEN = 511

_vp = C.c_void_p


class Config(C.Structure):
    _fields_ = [("bin_size", C.c_float), ("n_top", C.c_uint32), ("mod_group", C.c_char_p),
                ("mod_mass", C.c_float), ("mz_error", C.c_float), ("fragment_types", C.c_char_p),
                ("device", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("n_psm", C.c_uint64), ("peak_off", _vp), ("pep", _vp), ("pep_off", _vp),
                ("n_of_mod", _vp), ("max_charge", _vp), ("aux_pos", _vp), ("aux_mass", _vp),
                ("aux_off", _vp)]


class Results(C.Structure):
    _fields_ = [("max_k", C.c_uint32), ("best_score", _vp), ("best_sig", _vp), ("n_sig", _vp),
                ("ascores", _vp), ("alt_mask", _vp)]


# every symbol include/pyascore_hip.h declares
SYMBOLS = {
    "pya_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "pya_destroy": (None, [_vp]),
    "pya_add_neutral_loss": (C.c_int, [_vp, C.c_char_p, C.c_float]),
    "pya_reload_env": (C.c_int, [_vp]),
    "pya_set_debug": (C.c_int, [_vp, C.c_char_p, C.c_char_p]),          # include/pyascore_debug.h (test-only)
    "pya_debug_wave_ops": (C.c_int, [_vp, _vp, _vp]),         # (test-only)
    "pya_score_one": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, C.c_uint64, C.c_int32, C.c_int32, _vp, _vp, C.c_uint64,
                                C.c_uint32, C.POINTER(Results)]),
    "pya_rescore_last_keep": (C.c_int, [_vp]),
    "pya_last_error": (C.c_char_p, [_vp]),
    "pya_error_index": (C.c_int64, [_vp]),
    "pya_score_batch": (C.c_int, [_vp, C.POINTER(Batch), _vp, _vp, C.c_uint32, C.POINTER(Results)]),
    "pya_set_workspace_budget": (C.c_int, [_vp, C.c_uint64]),
    "pya_get_workspace_budget": (C.c_uint64, [_vp]),
    "pya_last_batch_status": (C.c_int, [_vp, _vp, C.c_uint64]),
    "pya_plan_create": (C.c_int, [_vp, C.POINTER(Batch), C.c_uint32, C.POINTER(_vp)]),
    "pya_plan_run": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(Results)]),
    "pya_plan_timings": (C.c_int, [_vp, C.POINTER(C.c_float * 4)]),
    "pya_one_times": (C.c_int, [_vp, C.POINTER(C.c_double * 12)]),
    "pya_plan_timings_sum": (C.c_int, [_vp, C.POINTER(C.c_double * 4), C.POINTER(C.c_uint32)]),
    "pya_plan_check": (C.c_int, [_vp]),
    "pya_pack_records": (C.c_int, [_vp, C.POINTER(Results), C.c_uint64, C.c_uint32, _vp, _vp]),
    "pya_plan_workspace_bytes": (C.c_uint64, [_vp]),
    "pya_plan_total_signatures": (C.c_uint64, [_vp]),
    "pya_plan_destroy": (None, [_vp]),
    "pya_get_pep_scores": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), _vp, _vp,
                                     _vp, _vp, _vp]),
    "pya_get_pep_scores_range": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pya_calculate_ambiguity": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _vp, C.c_float, C.c_uint64,
                                          _vp, C.c_float, C.POINTER(C.c_float)]),
    "pya_format_peptide": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int32, _vp, _vp, C.c_uint64,
                                     C.c_uint64, C.c_int32, C.c_char_p, C.c_uint64]),
    "pya_format_peptides": (C.c_int, [_vp, C.POINTER(Batch), C.c_uint64, _vp, _vp, _vp, _vp, _vp, C.c_uint64]),
    "pya_count_sites": (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_int32), _vp]),
    "pya_debug_sort": (C.c_int, [_vp, _vp, C.c_uint32, _vp]),
    "pya_version": (C.c_char_p, []),
    # include/pyascore_aux.h
    "pya_spectra_create": (_vp, [C.c_float, C.c_uint64]),
    "pya_spectra_destroy": (None, [_vp]),
    "pya_spectra_consume": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "pya_spectra_info": (None, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "pya_spectra_window_size": (C.c_int64, [_vp, C.c_uint64]),
    "pya_spectra_peak": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pya_modpep_create": (_vp, [C.c_char_p, C.c_float, C.c_float, C.c_char_p]),
    "pya_modpep_destroy": (None, [_vp]),
    "pya_modpep_last_error": (C.c_char_p, [_vp]),
    "pya_modpep_add_neutral_loss": (C.c_int, [_vp, C.c_char_p, C.c_float]),
    "pya_modpep_consume_peptide": (C.c_int, [_vp, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, _vp, _vp, C.c_uint64]),
    "pya_modpep_n_modifiable": (C.c_int64, [_vp]),
    "pya_modpep_consume_peak": (C.c_int, [_vp, C.c_float, C.c_uint64]),
    "pya_modpep_get_match": (C.c_int, [_vp, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]),
    "pya_modpep_get_peptide": (C.c_int64, [_vp, _vp, C.c_uint64, C.c_char_p, C.c_uint64]),
    "pya_modpep_site_ions": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_char, C.c_uint64, _vp, C.c_uint64,
                                       C.POINTER(C.c_uint64), _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "pya_fgraph_create": (_vp, [_vp, C.c_char, C.c_uint64]),
    "pya_fgraph_destroy": (None, [_vp]),
    "pya_fgraph_type": (C.c_char, [_vp]),
    "pya_fgraph_charge": (C.c_uint64, [_vp]),
    "pya_fgraph_reset_iterator": (C.c_int, [_vp]),
    "pya_fgraph_incr_signature": (C.c_int, [_vp]),
    "pya_fgraph_is_signature_end": (C.c_int, [_vp]),
    "pya_fgraph_reset_fragment": (C.c_int, [_vp]),
    "pya_fgraph_incr_fragment": (C.c_int, [_vp]),
    "pya_fgraph_is_fragment_end": (C.c_int, [_vp]),
    "pya_fgraph_is_loss": (C.c_int, [_vp]),
    "pya_fgraph_set_signature": (C.c_int, [_vp, _vp, C.c_uint64]),
    "pya_fgraph_get_signature": (C.c_int64, [_vp, _vp, C.c_uint64]),
    "pya_fgraph_fragment_mz": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "pya_fgraph_fragment_size": (C.c_uint64, [_vp]),
    "pya_fgraph_fragment_seq": (C.c_int64, [_vp, C.c_char_p, C.c_uint64]),
    "pya_log_sum": (C.c_float, [C.c_float, C.c_float]),
    "pya_log_bin_coef": (C.c_int, [C.c_uint64, C.c_uint64, C.POINTER(C.c_float)]),
    "pya_binomial": (C.c_int, [C.c_float, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_float)]),
    "pya_power_set_sums": (C.c_int64, [_vp, C.c_uint64, C.c_uint64, _vp, C.c_uint64]),
}

_lib = None


def _share_torch_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch-ROCm ships its own copy (torch/lib/libamdhip64.so,
    soname libamdhip64.so.7) and asks for it by FILE name; this library asks for the soname.  Loaded
    torch-first, the loader hands us torch's copy; loaded the other way round it would bring in a second
    runtime next to /opt/rocm's, torch would then find "No HIP GPUs", and device.DevicePlan could not
    take torch's device pointers.  So when a ROCm torch is installed, its copy goes in first (without
    importing torch)."""
    if any("libamdhip64" in line for line in open("/proc/self/maps")):
        return                                    # a runtime is already in the process: the loader reuses it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Loads the HIP library (no device needed for loading; pya_create needs one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyascore_amd: %s is missing. Build it with `python -m pyascore_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        if os.environ.get("PYA_LIB_OLD") and not hasattr(lib, name):
            continue                     # (A/B against an older build of the library: bench.py only)
        fn = getattr(lib, name)          # AttributeError = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
