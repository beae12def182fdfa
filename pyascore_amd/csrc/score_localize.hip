/* score_localize.hip -- kernel 2+3 fused: PepScores, winner and per-site Ascores of one PSM in one
 * pass of one wavefront, for PSMs with few site assignments and the plain scorer settings (no
 * neutral losses, fragment charge 1, one ion type per direction).  See fused_core.hip.h.
 *
 * HBM traffic per PSM: the retained table (5 B x R), the peptide, its descriptor and its signature
 * table (L2 resident) in; the 64-byte result out.  No per-signature scores, count records or m/z
 * grid go through HBM (they do on the score_signatures -> rank_and_localize route: 28 B x C(n,k) +
 * 512 B written and read back per PSM).  PSMs this kernel cannot finish are appended to a
 * hand-over list and redone by the general localize instantiation (pya_launch_localize_redo).
 */
#include "fused_core.hip.h"

#ifndef FUSED_WAVES
#define FUSED_WAVES 6
#endif

template <bool BOTH, bool ZM, bool WIDE = false>
__global__ __launch_bounds__(64, FUSED_WAVES) void pya_score_localize_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
    uint32_t ent_cap, uint32_t push_cap, uint32_t *redo_count, uint32_t *redo_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const bool declined = fused_body<BOTH, ZM, WIDE>(b, psm, lds_raw, cap, n_cap, stride, pos_cap, ent_cap, push_cap);
    if (declined && lane_id() == 0) redo_ids[atomicAdd(redo_count, 1u)] = psm;
}

extern "C" size_t pya_fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap,
                                      uint32_t push_cap, uint32_t both, uint32_t multi_z) {
    return fused_lds_bytes(cap, n_cap, stride, pos_cap, ent_cap, push_cap, both ? 2u : 1u, multi_z != 0);
}

extern "C" int pya_launch_fused(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap,
                                uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap, uint32_t both,
                                uint32_t multi_z, uint32_t *d_redo_count, uint32_t *d_redo_ids, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = fused_lds_bytes(cap, n_cap, stride, pos_cap, ent_cap, push_cap, both ? 2u : 1u, multi_z != 0);
#define PYA_FUSED_LAUNCH(B, Z, W)                                                                                          \
    do {                                                                                                                   \
        hipError_t e = PYA_ENSURE_MAX_LDS((pya_score_localize_kernel<B, Z, W>));                                           \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((pya_score_localize_kernel<B, Z, W>), dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, \
                           n_cap, stride, pos_cap, ent_cap, push_cap, d_redo_count, d_redo_ids);                           \
    } while (0)
    /* (both directions and more than 32 site assignments in the launch: the instantiation that walks them in two passes) */
    const bool wide = both && n_cap > 32u;
    if (both && multi_z && wide) PYA_FUSED_LAUNCH(true, true, true);
    else if (both && wide) PYA_FUSED_LAUNCH(true, false, true);
    else if (both && multi_z) PYA_FUSED_LAUNCH(true, true, false);
    else if (both) PYA_FUSED_LAUNCH(true, false, false);
    else if (multi_z) PYA_FUSED_LAUNCH(false, true, false);
    else PYA_FUSED_LAUNCH(false, false, false);
#undef PYA_FUSED_LAUNCH
    return (int)hipGetLastError();
}
