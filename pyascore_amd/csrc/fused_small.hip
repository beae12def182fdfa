/* fused_small.hip -- scoring + localisation in ONE kernel for PSMs with C(n,k) <= 64 site
 * assignments (the bulk of real batches and BASELINE cfg1/cfg2/most of cfg3): one PSM per
 * wavefront, retained-peak table in (from bin_spectra), summary record out.
 *
 *   walk              one (signature, direction) walker per lane -> PepScore of every signature
 *                     (depth scores and weighted score stay in the lane's registers)
 *   sort emulation    front element of the reference's std::sort -> winner
 *   loc_ascore_all    per-site Ascores from the site-determining ions of the tied best
 *                     single-move competitors
 *
 * Same device code as score_signatures + localize (walk_core / localize_core headers); what the
 * fusion removes is a kernel boundary, the HBM round trip of the weighted scores, the second
 * staging of the retained table / grid / residues and the recomputation of the winner's and the
 * competitors' depth scores.  Binning stays a separate kernel: it runs at 32 waves per CU, which a
 * fused kernel with this one's registers and LDS cannot (measured, profiles/r01_b).
 */
#include "walk_core.hip.h"
#include "localize_core.hip.h"

#define FUSED_MAX_SIG 64

#define FUSED_PUSHED 64          /* single-move competitors of a winner among <= 64 signatures */

struct FusedLds {
    uint16_t *nl_present;   /* [256] */
    float *nl_uniq;         /* [PYA_MAX_UNIQ] */
    uint16_t *grid;         /* [PYA_GRID_CELLS] */
    PushedEntry *pushed;    /* [FUSED_PUSHED] */
    uint32_t *site_max;     /* [64] */
    uint32_t *n_pushed;     /* [4] */
    float *ws_all;          /* [FUSED_MAX_SIG] weighted score per signature (pre-sort order) */
    float *scores_all;      /* [FUSED_MAX_SIG * 10] */
    PeakEntry *t_e;         /* [peak_cap + PYA_TABLE_PAD] */
    unsigned char *scratch; /* sort arrays, later the localisation work area */
};

extern "C" size_t pya_fused_lds_bytes(uint32_t peak_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb) {
    size_t fixed = 512 + PYA_MAX_UNIQ * 4 + PYA_GRID_CELLS * 2 + FUSED_PUSHED * 16 + 64 * 4 + 16 +
                   FUSED_MAX_SIG * 4 + FUSED_MAX_SIG * 10 * 4;
    size_t table = ((size_t)peak_cap + PYA_TABLE_PAD) * 8;
    size_t srt = (size_t)FUSED_MAX_SIG * 10 + 64;
    size_t loc = pya_loc_lds_bytes(pos_cap, pool_cap, sb);
    return fixed + table + (srt > loc ? srt : loc) + 64;
}

__global__ __launch_bounds__(64, 5) void pya_fused_small_kernel(BatchDev b, const uint32_t *psm_ids,
                                                             uint32_t n_ids, uint32_t peak_cap,
                                                             uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                                                             uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[blockIdx.x];
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const int k = b.n_of_mod[psm];
    const uint32_t max_k = b.max_k;
    float *out_asc = b.ascores + (size_t)psm * max_k;
    uint64_t *out_alt = b.alt_mask + (size_t)psm * max_k;
    for (uint32_t a = lane; a < max_k; a += 64) {
        out_asc[a] = 0.f;
        out_alt[a] = 0ull;
    }

    FusedLds f;
    f.nl_present = (uint16_t *)lds_raw;
    f.grid = f.nl_present + 256;
    f.nl_uniq = (float *)(f.grid + PYA_GRID_CELLS);
    f.pushed = (PushedEntry *)(f.nl_uniq + PYA_MAX_UNIQ);
    f.site_max = (uint32_t *)(f.pushed + FUSED_PUSHED);
    f.n_pushed = f.site_max + 64;
    f.ws_all = (float *)(f.n_pushed + 4);
    f.scores_all = f.ws_all + FUSED_MAX_SIG;
    f.t_e = (PeakEntry *)(f.scores_all + FUSED_MAX_SIG * 10);
    f.scratch = (unsigned char *)(f.t_e + peak_cap + PYA_TABLE_PAD);

    if (b.status[psm] != PYA_ST_OK) {
        if (lane == 0) {
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return;
    }

    /* ---- 1. stage the retained-peak table and its m/z grid ---- */
    STAMP_BEGIN();
    LocCtx ctx;
    ctx.b = &b;
    ctx.cfg = cfg;
    stage_peak_table(b, psm, f.t_e, &ctx.tab);
    ctx.nl.n_nl = cfg->n_nl;
    ctx.nl.present = f.nl_present;
    ctx.nl.uniq = f.nl_uniq;
    if (ctx.nl.n_nl) {
        for (int i = lane; i < 256; i += 64) f.nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) f.nl_uniq[lane] = cfg->uniq[lane];
    }
    if (lane == 0) *f.n_pushed = 0;
    f.site_max[lane] = 0;
    wave_lds_sync();
    grid_build(&ctx.tab, f.grid);
    unsigned char *scratch = f.scratch;
    STAMP(b, 1);

    /* ---- 2. PepScore of every signature ---- */
    const Residues res = load_residues(b, cfg, psm);
    const int N = (int)b.n_sig[psm];
    const int n_sites = __popcll(res.site_mask);
    const uint64_t *order = b.order_tab + b.order_off[psm];
    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = ctx.nl.n_nl;
    env.nl_present = f.nl_present;
    env.nl_uniq = f.nl_uniq;
    env.L = res.L;
    env.zmax = b.max_charge[psm];
    wave_lds_sync();
    STAMP(b, 2);

    int fail = 0;
    float my_ws = -1.f;
    uint64_t my_bits = 0ull;
    if (N > 0) {
        const bool both_dirs = cfg->n_fwd > 0 && cfg->n_fwd < cfg->n_types;
        const bool split = N <= 32 && both_dirs;
        const int s = split ? (lane & 31) : lane;
        const bool active = s < N;
        my_bits = active ? order[s] : 0ull;
        const uint64_t resmask = deposit_sites(my_bits, res.site_mask);
        Hist h = {0ull, 0ull, 0ull};
        uint32_t nfrag = 0;
        if (walk_is_simple(env)) {
            if (split) {
                walk_simple(env, res, ctx.tab, resmask, lane >> 5, active, h, nfrag);
                fold_upper_half(h, nfrag);
            } else {
                if (cfg->n_fwd > 0) walk_simple(env, res, ctx.tab, resmask, 0, active, h, nfrag);
                if (cfg->n_fwd < cfg->n_types) walk_simple(env, res, ctx.tab, resmask, 1, active, h, nfrag);
            }
        } else if (split) {
            walk(env, res, ctx.tab, resmask, lane >> 5, active, h, nfrag);
            fold_upper_half(h, nfrag);
        } else {
            if (cfg->n_fwd > 0) walk(env, res, ctx.tab, resmask, 0, active, h, nfrag);
            if (cfg->n_fwd < cfg->n_types) walk(env, res, ctx.tab, resmask, 1, active, h, nfrag);
        }
        if (active && (!split || lane < 32)) {
            uint32_t acc = 0;
            double sum = 0.;
            const bool ok = nfrag <= b.lut_n_max;
            const uint32_t off = ok ? lut_row(nfrag) : 0u;
            if (!ok) fail = 1;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                acc += hist_get(h, d);
                const float sc = ok ? b.lut[off + (uint32_t)d * (nfrag + 1) + acc] : 0.f;
                f.scores_all[s * 10 + d] = sc;
                const float prod = cfg->weights[d] * sc;                 /* float product ...   */
                sum = sum + (double)prod;                                /* ... double sum      */
            }
            my_ws = ok ? (float)sum : -1.f;
            f.ws_all[s] = my_ws;
        }
    }
    wave_lds_sync();
    STAMP(b, 3);

    /* Ascore::isUnambiguous, cpp/Ascore.cpp:38-51 */
    if (k >= n_sites) {
        for (int a = lane; a < k && a < (int)max_k; a += 64) out_asc[a] = __builtin_huge_valf();
        const bool any_fail = __any(fail != 0);
        if (lane == 0) {
            b.best_score[psm] = N > 0 ? f.ws_all[0] : -1.f;
            b.best_sig[psm] = N > 0 ? order[0] : 0ull;
            b.n_sig_out[psm] = N;
            if (any_fail) b.status[psm] = PYA_ST_LUT_RANGE;
        }
        return;
    }

    /* ---- 3. winner = front of the reference's sort (cpp/Ascore.cpp:141-146) ---- */
    SortLds srt;
    srt.key = (float *)scratch;
    srt.idx = (uint16_t *)(srt.key + N);
    srt.lpos = srt.idx + N;
    srt.rpos = srt.lpos + N;
    if (lane < N) {
        srt.key[lane] = f.ws_all[lane];
        srt.idx[lane] = (uint16_t)lane;
    }
    wave_lds_sync();
    sort_introsort_loop(srt, N, true);
    uint32_t u = lane < N ? __float_as_uint(srt.key[lane]) : 0u;        /* scores >= 0 */
    const uint32_t kmax = wave_max_u32(u);
    const uint32_t first_pos = wave_min_u32((lane < N && u == kmax) ? (uint32_t)lane : 0xffffffffu);
    const uint32_t best_i = srt.idx[first_pos];
    const float best_ws = __uint_as_float(kmax);
    const uint64_t best_bits = order[best_i];
    wave_lds_sync();
    STAMP(b, 4);

    /* ---- 4. best single-move competitors per modified site (cpp/Ascore.cpp:212-254) ---- */
    {
        const bool in = lane < N;
        const uint64_t c = in ? order[lane] : 0ull;
        const uint64_t gone = best_bits & ~c, came = c & ~best_bits;
        const bool single = in && __popcll(gone) == 1 && __popcll(came) == 1;
        const int a = single ? __popcll(best_bits & (gone - 1)) : 0;
        const uint32_t wbits = in ? __float_as_uint(f.ws_all[lane]) : 0u;
        if (single) atomicMax(&f.site_max[a], wbits);
        wave_lds_sync();
        if (single && wbits == f.site_max[a]) {
            const uint32_t slot = atomicAdd(f.n_pushed, 1u);
            if (slot < FUSED_PUSHED) {
                PushedEntry pe;
                pe.bits = c;
                pe.ws = __uint_as_float(wbits);
                pe.idx = (uint32_t)lane;
                f.pushed[slot] = pe;
            }
        }
        wave_lds_sync();
    }
    const uint32_t np = *f.n_pushed;                    /* <= FUSED_PUSHED */
    STAMP(b, 5);

    /* ---- 5. Ascores ---- */
    ctx.w = loc_carve(scratch, pos_cap, pool_cap, sb);
    ctx.sb = (int)sb;
    ctx.gtp = (int)gtp;
    ctx.L = res.L;
    ctx.zmax = env.zmax;
    ctx.pos_cap = pos_cap;
    ctx.pool_cap = pool_cap;
    ctx.w.m0[lane] = res.m0;
    ctx.w.m1[lane] = res.m1;
    ctx.w.nlp[lane] = (uint8_t)res.nl;
    if (lane == 0) ctx.w.sig_mask[0] = deposit_sites(best_bits, res.site_mask);
    wave_lds_sync();
    float my_asc = __builtin_huge_valf();
    uint64_t my_alt = 0ull;
    loc_ascore_all(ctx, f.pushed, np, f.scores_all, nullptr, best_bits, best_ws, best_i,
                   res.site_mask, &my_asc, &my_alt, &fail);
    STAMP(b, 6);
    if (lane < k && lane < (int)max_k) {
        out_asc[lane] = my_asc;
        out_alt[lane] = my_alt;
    }
    const bool any_fail = __any(fail != 0);
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (any_fail) b.status[psm] = PYA_ST_LUT_RANGE;
    }
}

extern "C" int pya_launch_fused_small(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids,
                                      uint32_t peak_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                                      uint32_t gtp, hipStream_t stream) {
    if (n_ids == 0) return 0;
    size_t lds = pya_fused_lds_bytes(peak_cap, pos_cap, pool_cap, sb);
    hipError_t e = hipFuncSetAttribute((const void *)pya_fused_small_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_fused_small_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, peak_cap,
                       pos_cap, pool_cap, sb, gtp);
    return (int)hipGetLastError();
}
