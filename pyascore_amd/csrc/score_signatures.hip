/* score_signatures.hip -- kernel 2: PepScore of every site assignment (one PSM per wavefront).
 *
 * Replaces ModifiedPeptide::initializeFragments + the consumePeak match cache + the
 * FragmentGraph walk of Ascore::accumulateCounts and Ascore::calculateFullScores
 * (cpp/ModifiedPeptide.cpp:81-150, :326-609; cpp/Ascore.cpp:53-139).
 *
 * Mapping: one (signature, direction) walker per lane -- with C(n,k) <= 32 both directions of
 * every signature run side by side, otherwise a lane walks forward then backward and lanes
 * loop over the signatures.  A walker keeps the reference's float32 running sum in a register,
 * derives every (neutral-loss variant, ion type, charge) m/z in the reference's double/float
 * order, looks it up in the wave's LDS copy of the retained-peak table (m/z grid + short scan)
 * and bumps its column of an LDS rank histogram (ds_add_u32).  Per-residue masses are staged in
 * LDS once per direction of travel.  All three kernels of the path are VALU-issue bound
 * (profiles/r01_d), so the walker is written for instruction count: ~40 VALU per fragment.  The
 * binomial tail is a host-built table (float32 chain in the reference's order), so the device
 * does integer counting + table reads + the exactly-rounded weighted sum.
 *
 * HBM traffic per PSM: retained table (5 B x R) + peptide bytes + 8 B x C(n,k) signature
 * table (L2 resident, shared by all PSMs of a shape) in; 4 B x C(n,k) weighted scores out.
 */
#include "score_core.hip.h"

#ifndef SCORE_WAVES
#define SCORE_WAVES 6
#endif
template <bool PREFIX>
__global__ __launch_bounds__(64, SCORE_WAVES) void pya_score_signatures_kernel(BatchDev b, const uint32_t *psm_ids,
                                                                  uint32_t n_ids, uint32_t cap, uint32_t with_nl,
                                                                  uint32_t compact) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    score_body<PREFIX>(b, psm_ids[xcd_slot(blockIdx.x, n_ids)], lds_raw, cap, with_nl, compact);
}

/* general settings (neutral losses, several ion types per direction): one lookup set per distinct node of the
 * assignment tree (score_core.hip.h: score_nodes_dir), walkers for what does not fit */
/* (its time is inversely proportional to its occupancy -- 5.3 / 5.9 / 6.3 ms on cfg4 with 16 / 14 / 12 wavefronts per CU --
 * so its LDS is sized by the launch to fit five per SIMD, and the registers follow: 96 with a handful of spills) */
__global__ __launch_bounds__(64, 5) void pya_score_nodes_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids,
                                                                           uint32_t cap, uint32_t with_nl, uint32_t node_cap,
                                                                           uint32_t node_cols, uint32_t node_words, uint32_t res_cap,
                                                                           uint32_t nl_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    score_body<false, true>(b, psm_ids[xcd_slot(blockIdx.x, n_ids)], lds_raw, cap, with_nl, 0u, node_cap, node_cols, node_words,
                            res_cap, nl_cap);
}

extern "C" size_t pya_score_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact) {
    return score_lds_bytes(cap, prefix, with_nl, compact);
}
extern "C" size_t pya_score_node_lds_bytes(uint32_t cap, uint32_t with_nl, uint32_t node_cap, uint32_t node_cols, uint32_t node_words,
                                           uint32_t res_cap, uint32_t nl_cap) {
    return score_lds_bytes(cap, 0, with_nl, 0, node_cap, node_cols, node_words, res_cap, nl_cap);
}

/* node_cap != 0 (only without `prefix`): the shared-node route of score_core.hip.h with room for node_cap nodes per
 * direction, node_cols histogram columns and a shape table of node_words 64-bit words */
extern "C" int pya_launch_score(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                                uint32_t prefix, uint32_t with_nl, uint32_t compact, uint32_t node_cap, uint32_t node_cols,
                                uint32_t node_words, uint32_t res_cap, uint32_t nl_cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    /* spectra near the 8192-peak limit need more than the default 64 KB of dynamic LDS */
    if (!prefix) compact = 0;
    if (prefix) node_cap = 0;
    if (!node_cap) {
        res_cap = 64;
        nl_cap = 256;
    }
    const size_t lds = score_lds_bytes(cap, prefix, with_nl, compact, node_cap, node_cols, node_words, res_cap, nl_cap);
    if (node_cap) {
        hipError_t en = PYA_ENSURE_MAX_LDS(pya_score_nodes_kernel);
        if (en != hipSuccess) return (int)en;
        hipLaunchKernelGGL(pya_score_nodes_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, with_nl, node_cap,
                           node_cols, node_words, res_cap, nl_cap);
        return (int)hipGetLastError();
    }
    hipError_t e = prefix ? PYA_ENSURE_MAX_LDS(pya_score_signatures_kernel<true>)
                          : PYA_ENSURE_MAX_LDS(pya_score_signatures_kernel<false>);
    if (e != hipSuccess) return (int)e;
    if (prefix)
        hipLaunchKernelGGL(pya_score_signatures_kernel<true>, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids,
                           cap, with_nl, compact);
    else
        hipLaunchKernelGGL(pya_score_signatures_kernel<false>, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids,
                           cap, with_nl, compact);
    return (int)hipGetLastError();
}
