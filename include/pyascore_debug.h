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
 *     PYA_NO_LOC_HASH PYA_NO_NODES PYA_NO_CNT PYA_HOST_TIMING PYA_STAMPS PYA_SLOW_NULL_STREAM
 *   numbers: PYA_DEBUG (bit set, common.h) PYA_PLAIN_MIN PYA_BIG_MIN_N PYA_TINY_MAX PYA_SORT_ROOM_MAX PYA_SB PYA_GTP
 *     PYA_HASH_PP PYA_NODE_CAP PYA_CHUNK_MB PYA_WORKSPACE_MB */
int pya_set_debug(pya_handle *h, const char *key, const char *value);

#ifdef __cplusplus
}
#endif
#endif
