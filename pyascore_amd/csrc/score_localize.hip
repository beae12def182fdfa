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
#include "bin_core.hip.h"
#include "fused_core.hip.h"
#include "fused_pack.hip.h"

#ifndef FUSED_WAVES
#define FUSED_WAVES 6
#endif

template <bool BOTH, bool ZM>
__global__ __launch_bounds__(64, FUSED_WAVES) void pya_score_localize_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
    uint32_t ent_cap, uint32_t push_cap, uint32_t *redo_count, uint32_t *redo_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const bool declined = fused_body<BOTH, ZM>(b, psm, lds_raw, cap, n_cap, stride, pos_cap, ent_cap, push_cap);
    if (declined && lane_id() == 0) redo_ids[atomicAdd(redo_count, 1u)] = psm;
}

/* Binning, scoring and localisation of a PSM in ONE pass of one wavefront: bin_core leaves the retained table in
 * LDS and the fused body takes it from there (through registers: the two layouts overlap).  The table still goes to
 * the workspace (stores nobody waits for: the general localize body needs it should the PSM be handed over), but it
 * is not read back -- 136 MB per cfg2 step -- and one launch with its ramp-up and tail is gone.  Spectra the
 * common-case binning declines (peaks out of m/z order, equal intensities at a window's top) are appended to the
 * bin hand-over list and finished by pya_bin_exact_kernel + the list form of the fused kernel. */
template <bool BOTH, bool ZM>
__global__ __launch_bounds__(64, FUSED_WAVES) void pya_bin_score_localize_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
    uint32_t ent_cap, uint32_t push_cap, uint32_t *redo_count, uint32_t *redo_ids, uint32_t *binredo_count,
    uint32_t *binredo_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const float *r_mz;
    const uint8_t *r_rank;
    int status;
    const int R = bin_core<false>(b, psm, lds_raw, cap, &r_mz, &r_rank, &status);
    if (R == PYA_BIN_REDO || R > FUSED_LOCAL_CHUNKS * 64 - PYA_TABLE_PAD) {
        if (R != PYA_BIN_REDO) bin_store(b, psm, R, status, r_mz, r_rank);      /* (a very long table: through the workspace) */
        if (lane_id() == 0) {
            if (R == PYA_BIN_REDO) b.redo_ids[atomicAdd(b.redo_count, 1u)] = psm;
            binredo_ids[atomicAdd(binredo_count, 1u)] = psm;
        }
        return;
    }
    bin_store(b, psm, R, status, r_mz, r_rank);
    LocalTable lt = {r_mz, r_rank, R < 0 ? 0 : R, R < 0 ? status : PYA_ST_OK};
    const bool declined = fused_body<BOTH, ZM>(b, psm, lds_raw, cap, n_cap, stride, pos_cap, ent_cap, push_cap, &lt);
    if (declined && lane_id() == 0) redo_ids[atomicAdd(redo_count, 1u)] = psm;
}

/* the PSMs the packed kernel passed on (peak pool too small, a residue at or below two tolerances): a
 * small grid strides over the list, one PSM per wavefront as above */
template <bool BOTH, bool ZM>
__global__ __launch_bounds__(64, FUSED_WAVES) void pya_score_localize_list_kernel(
    BatchDev b, const uint32_t *count, const uint32_t *ids, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
    uint32_t ent_cap, uint32_t push_cap, uint32_t *redo_count, uint32_t *redo_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t n = *count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t psm = ids[k];
        const bool declined = fused_body<BOTH, ZM>(b, psm, lds_raw, cap, n_cap, stride, pos_cap, ent_cap, push_cap);
        if (declined && lane_id() == 0) redo_ids[atomicAdd(redo_count, 1u)] = psm;
        wave_lds_sync();
    }
}

/* several PSMs per wavefront (fused_pack.hip.h) */
#ifndef PACK_WAVES
#define PACK_WAVES 4
#endif
template <bool BOTH>
__global__ __launch_bounds__(64, PACK_WAVES) void pya_score_localize_pack_kernel(
    BatchDev b, const uint64_t *pdesc, uint32_t n_ids, uint32_t G, uint32_t pool_cap, uint32_t n_cap, uint32_t stride,
    uint32_t pos_cap, uint32_t push_cap, uint32_t kc, uint32_t *redo_count, uint32_t *redo_ids, uint32_t *over_count,
    uint32_t *over_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t nblk = (n_ids + G - 1) / G;
    if (blockIdx.x >= nblk) return;
    const uint32_t first = xcd_slot(blockIdx.x, nblk) * G;
    fused_pack_body<BOTH>(b, pdesc, n_ids, first, G, lds_raw, pool_cap, n_cap, stride, pos_cap, push_cap, kc, redo_count,
                          redo_ids, over_count, over_ids);
}

extern "C" size_t pya_pack_lds_bytes(uint32_t G, uint32_t pool_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t push_cap,
                                     uint32_t kc, uint32_t both) {
    const uint32_t ndir = both ? 2u : 1u;
    return pack_lds_bytes(G, pool_cap, n_cap, ndir * n_cap + 4u, pos_cap, push_cap, kc, ndir);
}

extern "C" int pya_launch_fused_pack(const BatchDev *b, const uint64_t *d_ids, uint32_t n_ids, uint32_t G, uint32_t pool_cap,
                                     uint32_t n_cap, uint32_t pos_cap, uint32_t push_cap, uint32_t kc, uint32_t both,
                                     uint32_t *d_redo_count, uint32_t *d_redo_ids, uint32_t *d_over_count, uint32_t *d_over_ids,
                                     hipStream_t stream) {
    if (n_ids == 0) return 0;
    const uint32_t ndir = both ? 2u : 1u, stride = ndir * n_cap + 4u;
    const size_t lds = pack_lds_bytes(G, pool_cap, n_cap, stride, pos_cap, push_cap, kc, ndir);
    const uint32_t nblk = (n_ids + G - 1) / G;
    hipError_t e = both ? PYA_ENSURE_MAX_LDS((pya_score_localize_pack_kernel<true>)) : PYA_ENSURE_MAX_LDS((pya_score_localize_pack_kernel<false>));
    if (e != hipSuccess) return (int)e;
    if (both)
        hipLaunchKernelGGL((pya_score_localize_pack_kernel<true>), dim3(nblk), dim3(64), lds, stream, *b, d_ids, n_ids, G, pool_cap,
                           n_cap, stride, pos_cap, push_cap, kc, d_redo_count, d_redo_ids, d_over_count, d_over_ids);
    else
        hipLaunchKernelGGL((pya_score_localize_pack_kernel<false>), dim3(nblk), dim3(64), lds, stream, *b, d_ids, n_ids, G, pool_cap,
                           n_cap, stride, pos_cap, push_cap, kc, d_redo_count, d_redo_ids, d_over_count, d_over_ids);
    return (int)hipGetLastError();
}

/* what the packed launches passed on, on the one-PSM-per-wavefront kernel (charge 1) */
extern "C" int pya_launch_fused_list(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max, uint32_t cap,
                                     uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap,
                                     uint32_t both, uint32_t multi_z, uint32_t *d_redo_count, uint32_t *d_redo_ids,
                                     hipStream_t stream) {
    if (n_max == 0) return 0;
    const size_t lds = fused_lds_bytes(cap, n_cap, stride, pos_cap, ent_cap, push_cap, both ? 2u : 1u, multi_z != 0);
    const uint32_t grid = n_max < 8192u ? n_max : 8192u;
#define PYA_FUSEDLIST_LAUNCH(B, Z)                                                                                      \
    do {                                                                                                                \
        hipError_t e = PYA_ENSURE_MAX_LDS((pya_score_localize_list_kernel<B, Z>));                                      \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL((pya_score_localize_list_kernel<B, Z>), dim3(grid), dim3(64), lds, stream, *b, d_count, d_ids, cap, \
                           n_cap, stride, pos_cap, ent_cap, push_cap, d_redo_count, d_redo_ids);                        \
    } while (0)
    if (both && multi_z) PYA_FUSEDLIST_LAUNCH(true, true);
    else if (both) PYA_FUSEDLIST_LAUNCH(true, false);
    else if (multi_z) PYA_FUSEDLIST_LAUNCH(false, true);
    else PYA_FUSEDLIST_LAUNCH(false, false);
#undef PYA_FUSEDLIST_LAUNCH
    return (int)hipGetLastError();
}

extern "C" size_t pya_bin_fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap,
                                          uint32_t push_cap, uint32_t both, uint32_t multi_z) {
    const size_t a = fused_lds_bytes(cap, n_cap, stride, pos_cap, ent_cap, push_cap, both ? 2u : 1u, multi_z != 0);
    const size_t bb = (((size_t)cap * 15 + 63) & ~(size_t)63) + 64;
    return a > bb ? a : bb;
}

extern "C" int pya_launch_bin_fused(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap,
                                    uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap, uint32_t both,
                                    uint32_t multi_z, uint32_t *d_redo_count, uint32_t *d_redo_ids, uint32_t *d_binredo,
                                    hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = pya_bin_fused_lds_bytes(cap, n_cap, stride, pos_cap, ent_cap, push_cap, both, multi_z);
#define PYA_BINFUSED_LAUNCH(B, Z)                                                                                       \
    do {                                                                                                                \
        hipError_t e = PYA_ENSURE_MAX_LDS((pya_bin_score_localize_kernel<B, Z>));                                       \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL((pya_bin_score_localize_kernel<B, Z>), dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, \
                           n_cap, stride, pos_cap, ent_cap, push_cap, d_redo_count, d_redo_ids, d_binredo, d_binredo + 64); \
    } while (0)
    if (both && multi_z) PYA_BINFUSED_LAUNCH(true, true);
    else if (both) PYA_BINFUSED_LAUNCH(true, false);
    else if (multi_z) PYA_BINFUSED_LAUNCH(false, true);
    else PYA_BINFUSED_LAUNCH(false, false);
#undef PYA_BINFUSED_LAUNCH
    return (int)hipGetLastError();
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
#define PYA_FUSED_LAUNCH(B, Z)                                                                                          \
    do {                                                                                                                \
        hipError_t e = PYA_ENSURE_MAX_LDS((pya_score_localize_kernel<B, Z>));                                           \
        if (e != hipSuccess) return (int)e;                                                                             \
        hipLaunchKernelGGL((pya_score_localize_kernel<B, Z>), dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, \
                           n_cap, stride, pos_cap, ent_cap, push_cap, d_redo_count, d_redo_ids);                        \
    } while (0)
    if (both && multi_z) PYA_FUSED_LAUNCH(true, true);
    else if (both) PYA_FUSED_LAUNCH(true, false);
    else if (multi_z) PYA_FUSED_LAUNCH(false, true);
    else PYA_FUSED_LAUNCH(false, false);
#undef PYA_FUSED_LAUNCH
    return (int)hipGetLastError();
}
