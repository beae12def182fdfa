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
#include "fused_core.hip.h"

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
    size_t a = PYA_BIN_WAVE_BYTES(cap);       /* (covers the exact body's 192 bytes of window starts too) */
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


/* ---------------------------------------------------------------------------------------------------------
 * PyAscore.score(): ONE PSM, lowest latency (Ascore.pyx:103-152 is called once per PSM by the reference's own
 * command line, pyascore/__main__.py:129-164).  No plan, no copies: the spectrum sits in pinned host memory
 * the device reads directly, the PSM's scalars and letters arrive in the kernel arguments, the results are
 * written straight into pinned host memory and a sequence number behind them tells the polling host thread
 * that they are complete.  One wavefront: bin -> (fused score + localize | score -> localize) back to back.
 * use_fused: 0 = score_body + the general localize body (also the retained, PYA_FLAG_KEEP form),
 *            1 = the fused body with ion types of both directions, 2 = of one direction; a PSM it hands over
 *            is finished by the general localize body right here.
 * ------------------------------------------------------------------------------------------------------- */
template <bool ZM>
__global__ __launch_bounds__(64) void pya_one_kernel(BatchDev b, OneMeta m, uint32_t cap, uint32_t prefix, uint32_t with_nl,
                                                     uint32_t compact, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap,
                                                     uint32_t sb, uint32_t gtp, uint32_t use_fused, uint32_t f_n_cap,
                                                     uint32_t f_stride, uint32_t f_ent_cap, uint32_t f_push_cap,
                                                     int32_t *host_status, uint32_t *host_flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    /* (the 100 MHz clock at the stage boundaries, handed to the host with the status: pya_one_times) */
    const uint64_t tk0 = __builtin_amdgcn_s_memrealtime(), cy0 = __builtin_amdgcn_s_memtime();
    /* the spectrum sits in host memory: the loads of its first 384 peaks go out before anything else and travel while
     * the scalars below are put into place (the peak count is a kernel argument, the spectrum starts at offset 0) */
    BinPre pre;
    bin_preload(b.mz, b.inten, m.n_peaks, &pre);
    /* the PSM's scalars into the batch arrays (all of them arrays of one PSM at offset 0) */
    if (lane == 0) {
        int64_t *w;
        w = (int64_t *)b.peak_off; w[0] = 0; w[1] = (int64_t)m.n_peaks;
        w = (int64_t *)b.pep_off;  w[0] = 0; w[1] = (int64_t)m.L;
        w = (int64_t *)b.aux_off;  w[0] = 0; w[1] = (int64_t)m.n_aux;
        w = (int64_t *)b.sig_off;  w[0] = 0; w[1] = (int64_t)m.n_sig;
        w = (int64_t *)b.ret_off;  w[0] = 0;
        ((int32_t *)b.n_of_mod)[0] = m.n_of_mod;
        ((int32_t *)b.max_charge)[0] = m.max_charge;
        ((uint8_t *)b.n_sites)[0] = (uint8_t)m.n_sites;
        ((uint32_t *)b.n_sig)[0] = m.n_sig;
        ((uint32_t *)b.order_off)[0] = m.order_off;
        b.ws_top[1] = 0u;
        *b.redo_count = 0u;
    }
    if (lane < 6) ((uint64_t *)b.desc)[lane] = m.desc[lane];
    if (lane < PYA_MAX_L / 8) ((uint64_t *)b.pep)[lane] = m.pep[lane];
    if (lane < PYA_ONE_MAX_AUX) {
        ((uint32_t *)b.aux_pos)[lane] = m.aux_pos[lane];
        ((float *)b.aux_mass)[lane] = m.aux_mass[lane];
    }
    __threadfence();
    wave_lds_sync();
    const uint64_t tk1 = __builtin_amdgcn_s_memrealtime();
    const float *r_mz;
    const uint8_t *r_rank;
    int bin_status;
    int R = bin_fast<false, true>(b, 0, lds_raw, cap, &r_mz, &r_rank, &bin_status, &pre);
    if (R == PYA_BIN_REDO) {
        wave_lds_sync();
        R = bin_core<true>(b, 0, lds_raw, cap, &r_mz, &r_rank, &bin_status);
    }
    bin_store(b, 0, R, bin_status, r_mz, r_rank);            /* (the general bodies and a retained PSM read it from the workspace) */
    const uint64_t tk2 = __builtin_amdgcn_s_memrealtime();
    bool general = use_fused == 0;
    if (!general && R <= FUSED_LOCAL_CHUNKS * 64 - PYA_TABLE_PAD) {
        /* the fused body takes the table bin_core left in LDS: no round trip through the workspace */
        LocalTable lt = {r_mz, r_rank, R < 0 ? 0 : R, R < 0 ? bin_status : PYA_ST_OK};
        const bool declined = use_fused == 1 ? fused_body<true, ZM>(b, 0, lds_raw, cap, f_n_cap, f_stride, pos_cap, f_ent_cap, f_push_cap, &lt)
                                             : fused_body<false, ZM>(b, 0, lds_raw, cap, f_n_cap, f_stride, pos_cap, f_ent_cap, f_push_cap, &lt);
        general = declined;                                  /* (wave-uniform) what it handed over is finished below */
    } else {
        __threadfence();
        wave_lds_sync();
        if (!general) {
            const bool declined = use_fused == 1 ? fused_body<true, ZM>(b, 0, lds_raw, cap, f_n_cap, f_stride, pos_cap, f_ent_cap, f_push_cap)
                                                 : fused_body<false, ZM>(b, 0, lds_raw, cap, f_n_cap, f_stride, pos_cap, f_ent_cap, f_push_cap);
            general = declined;
        } else {
            if (prefix) score_body<true>(b, 0, lds_raw, cap, with_nl, compact);
            else score_body<false>(b, 0, lds_raw, cap, with_nl, 0u);
        }
    }
    const uint64_t tk3 = __builtin_amdgcn_s_memrealtime();
    if (general) {
        __threadfence();
        wave_lds_sync();
        localize_body<false>(b, 0, lds_raw, push_cap, pos_cap, pool_cap, sb, gtp);
    }
    /* results are in host memory (the output arrays point there), written by several lanes: every lane makes its own
     * stores visible to the host (system scope), then lane 0 writes the status, an echo of the sequence number next
     * to it and -- after another system fence -- the sequence number the host polls */
    __threadfence_system();
    wave_lds_sync();
    if (lane == 0) {
        host_status[0] = b.status[0];
        host_status[1] = (int32_t)m.seq;
        host_status[2] = (int32_t)(tk1 - tk0);
        host_status[3] = (int32_t)(tk2 - tk1);
        host_status[4] = (int32_t)(tk3 - tk2);
        host_status[5] = (int32_t)(__builtin_amdgcn_s_memrealtime() - tk3);
        host_status[6] = (int32_t)(__builtin_amdgcn_s_memtime() - cy0);       /* shader clock cycles of the whole kernel */
        __threadfence_system();
        __hip_atomic_store(host_flag, m.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" size_t pya_one_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact, uint32_t push_cap,
                                    uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t use_fused,
                                    uint32_t f_n_cap, uint32_t f_stride, uint32_t f_ent_cap, uint32_t f_push_cap, uint32_t multi_z) {
    size_t a = pya_tiny_lds_bytes(cap, prefix, with_nl, compact, push_cap, n_cap, pos_cap, pool_cap, sb);
    if (use_fused) {
        const size_t fb = fused_lds_bytes(cap, f_n_cap, f_stride, pos_cap, f_ent_cap, f_push_cap, use_fused == 1 ? 2u : 1u, multi_z != 0);
        a = a > fb ? a : fb;
    }
    return a;
}

extern "C" int pya_launch_one(const BatchDev *b, const OneMeta *m, uint32_t cap, uint32_t prefix, uint32_t with_nl,
                              uint32_t compact, uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap,
                              uint32_t sb, uint32_t gtp, uint32_t use_fused, uint32_t f_n_cap, uint32_t f_stride,
                              uint32_t f_ent_cap, uint32_t f_push_cap, uint32_t multi_z, int32_t *host_status,
                              uint32_t *host_flag, hipStream_t stream) {
    const size_t lds = pya_one_lds_bytes(cap, prefix, with_nl, compact, push_cap, n_cap, pos_cap, pool_cap, sb, use_fused, f_n_cap,
                                         f_stride, f_ent_cap, f_push_cap, multi_z);
    hipError_t e = multi_z ? PYA_ENSURE_MAX_LDS(pya_one_kernel<true>) : PYA_ENSURE_MAX_LDS(pya_one_kernel<false>);
    if (e != hipSuccess) return (int)e;
    if (multi_z)
        hipLaunchKernelGGL(pya_one_kernel<true>, dim3(1), dim3(64), lds, stream, *b, *m, cap, prefix, with_nl, compact, push_cap,
                           pos_cap, pool_cap, sb, gtp, use_fused, f_n_cap, f_stride, f_ent_cap, f_push_cap, host_status, host_flag);
    else
        hipLaunchKernelGGL(pya_one_kernel<false>, dim3(1), dim3(64), lds, stream, *b, *m, cap, prefix, with_nl, compact, push_cap,
                           pos_cap, pool_cap, sb, gtp, use_fused, f_n_cap, f_stride, f_ent_cap, f_push_cap, host_status, host_flag);
    return (int)hipGetLastError();
}
