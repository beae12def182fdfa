/* tiny_batch.hip -- the whole path of a PSM in ONE launch, for batches of a handful of PSMs.
 *
 * PyAscore.score() scores one PSM per call (Ascore.pyx:103-152 has the same shape), and a batch of
 * one is launch-bound: bin_spectra, its exact variant, score_signatures and localize are four or
 * five dependent launches of one wavefront each, ~10 us apiece.  Here one wavefront per PSM runs
 * the same three bodies back to back (bin_core -> score_body -> localize_body, general
 * instantiations; the hand-over between them still goes through the workspace in global memory),
 * which takes the per-call device time from ~45 us to ~25 us.  Occupancy does not matter for a
 * handful of wavefronts, which is why this is NOT how big batches run (dead ends).
 */
#include "bin_core.hip.h"
#include "score_core.hip.h"
#include "localize_body.hip.h"

__global__ __launch_bounds__(64) void pya_tiny_batch_kernel(BatchDev b, uint32_t n_psm, uint32_t cap, uint32_t prefix,
                                                            uint32_t with_nl, uint32_t compact, uint32_t push_cap,
                                                            uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                                                            uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t psm = blockIdx.x;
    if (psm >= n_psm) return;
    {
        const float *r_mz;
        const uint8_t *r_rank;
        int status;
        int R = bin_core<false>(b, psm, lds_raw, cap, &r_mz, &r_rank, &status);
        if (R == PYA_BIN_REDO) {
            wave_lds_sync();
            R = bin_core<true>(b, psm, lds_raw, cap, &r_mz, &r_rank, &status);
        }
        bin_store(b, psm, R, status, r_mz, r_rank);
    }
    /* the next stage reads what this wavefront just wrote to global memory */
    __threadfence();
    wave_lds_sync();
    if (prefix) score_body<true>(b, psm, lds_raw, cap, with_nl, compact);
    else score_body<false>(b, psm, lds_raw, cap, with_nl, 0u);
    __threadfence();
    wave_lds_sync();
    localize_body<false>(b, psm, lds_raw, push_cap, pos_cap, pool_cap, sb, gtp);
}

extern "C" size_t pya_tiny_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact,
                                     uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap,
                                     uint32_t sb) {
    size_t a = (((size_t)cap * 15 + 63) & ~(size_t)63) + 192;
    size_t s = score_lds_bytes(cap, prefix, with_nl, prefix ? compact : 0u);
    size_t l = localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
    a = a > s ? a : s;
    return a > l ? a : l;
}

extern "C" int pya_launch_tiny(const BatchDev *b, uint32_t n_psm, uint32_t cap, uint32_t prefix, uint32_t with_nl,
                               uint32_t compact, uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap,
                               uint32_t pool_cap, uint32_t sb, uint32_t gtp, hipStream_t stream) {
    if (n_psm == 0) return 0;
    const size_t lds = pya_tiny_lds_bytes(cap, prefix, with_nl, compact, push_cap, n_cap, pos_cap, pool_cap, sb);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_tiny_batch_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_tiny_batch_kernel, dim3(n_psm), dim3(64), lds, stream, *b, n_psm, cap, prefix, with_nl,
                       compact, push_cap, pos_cap, pool_cap, sb, gtp);
    return (int)hipGetLastError();
}
