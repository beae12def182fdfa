/* pyascore_debug.h -- TEST-ONLY part of libpyascore_hip.so's C ABI.
 *
 * The kernels behind pyascore.PyAscore.score are several routes that must all give the reference's results
 * (fused / lean / general localisation, shared-node or per-walker scoring, the sort emulation, the hand-over lists
 * between kernels ...).  The parity suite drives every one of them on the same inputs; which route a PSM takes in
 * production is decided by its shape alone.  The switches that force a route, make a kernel decline what it would
 * take, or resize a table so that a hand-over happens, are set PER HANDLE through this call and through nothing else:
 * no environment variable selects a route (the library reads four variables, none of them a route: pyascore_hip.h,
 * pya_reload_env).  No reference counterpart: the reference has one route.
 */
#ifndef PYASCORE_DEBUG_H
#define PYASCORE_DEBUG_H
#include "pyascore_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Sets one debug switch of the handle.  `key` is the switch's name, `value` its value as text; value == NULL puts the
 * switch back to its production default.  pya_reload_env resets all of them.  Unknown names: PYA_ERR_ARG.
 *   flags (any non-NULL value = on): PYA_NO_PLAIN PYA_NO_FUSED PYA_NO_BIG PYA_NO_TINY PYA_NO_PREFIX PYA_NO_CHUNKS
 *     PYA_NO_UPLOAD_THREAD PYA_ONE_PEAK_CLASS PYA_PEAK_CLASSES PYA_ONE_LDS_CLASS PYA_SORT_ROOM PYA_NO_BIG_INLINE
 *     PYA_NO_LOC_HASH PYA_NO_NODES PYA_NO_CNT PYA_HOST_TIMING PYA_STAMPS PYA_SLOW_NULL_STREAM PYA_NO_FORK (r06: the
 *     fused family behind the scoring kernels on the caller's stream instead of beside them on the plan's side stream)
 *   numbers: PYA_DEBUG (bit set, common.h) PYA_PLAIN_MIN PYA_BIG_MIN_N PYA_TINY_MAX PYA_SORT_ROOM_MAX PYA_SB PYA_GTP
 *     PYA_HASH_PP PYA_NODE_CAP PYA_CHUNK_MB PYA_WORKSPACE_MB
 *     PYA_BIN_SELECT_MIN (r06: peak classes above this many peaks are binned by selection, csrc/bin_select.hip.h; default 640,
 *     0 = every class, a huge value = none) PYA_BIN_SELECT_SCAP (survivor slots per spectrum there; default 768) */
int pya_set_debug(pya_handle *h, const char *key, const char *value);

/* The wavefront primitives every kernel leans on (csrc/device_common.hip.h: prefix sums and reductions by DPP, the rank of
 * a lane in a ballot), run by one full wavefront on 64 values (as every call site does):
 *   out[0..63]    exclusive prefix sum of in[]
 *   out[64..127]  inclusive prefix sums within each half of 32 lanes
 *   out[128..191] inclusive prefix sum over the 64 lanes
 *   out[192..255] rank of the lane among the lanes whose value is odd
 *   out[256..262] total, sum, max and min (as unsigned), max and min of the values read as floats, min rank of an odd value's lane
 * No reference counterpart. */
int pya_debug_wave_ops(pya_handle *h, const int32_t in[64], int32_t out[263]);

#ifdef __cplusplus
}
#endif
#endif
