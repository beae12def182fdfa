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
 * and bumps a packed rank histogram.  Per-residue data sits in one register per lane and is
 * broadcast with v_readlane (no LDS).  The binomial tail is a host-built table (float32 chain in
 * the reference's order), so the device does integer counting + table reads + the
 * exactly-rounded weighted sum.
 *
 * HBM traffic per PSM: retained table (5 B x R) + peptide bytes + 8 B x C(n,k) signature
 * table (L2 resident, shared by all PSMs of a shape) in; 4 B x C(n,k) weighted scores out.
 */
#include "device_common.hip.h"

DEV float lut_score(const BatchDev &b, uint32_t depth, uint32_t k, uint32_t n) {
    return b.lut[lut_row(n) + depth * (n + 1) + k];
}

#include "walk_core.hip.h"

/* With many site assignments most of them agree on the first modifiable residues of a
 * direction, and a fragment's m/z depends only on the pattern of the residues it contains.  So
 * the first PREFIX_SITES sites of each direction are walked once per PATTERN (2^6 = 64 patterns,
 * one per lane) and every signature resumes from its pattern's state (float32 running sum,
 * neutral-loss stack, rank histogram) -- bit-identical to walking from the start, because it IS
 * the same sequence of float additions.  This is the first level of the reference's prefix-
 * sharing fragment tree (cpp/Ascore.cpp:69-109) laid out for a wavefront. */
#define PREFIX_SITES 6
struct PrefixState {
    float running;
    uint32_t nl_state;
    uint32_t nfrag;
    uint32_t pad;
    uint64_t ha, hb, hc;
};

/* PREFIX is a template parameter so that the small-C(n,k) instantiation does not carry the
 * registers of the shared-prefix path (60 vs 77 VGPRs = 8 vs 6 waves per SIMD). */
template <bool PREFIX>
__global__ __launch_bounds__(64) void pya_score_signatures_kernel(BatchDev b, const uint32_t *psm_ids,
                                                                  uint32_t n_ids, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[blockIdx.x];
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;

    uint16_t *nl_present = (uint16_t *)lds_raw;             /* [256] */
    uint16_t *grid = nl_present + 256;                      /* [PYA_GRID_CELLS] */
    float *nl_uniq = (float *)(grid + PYA_GRID_CELLS);      /* [PYA_MAX_UNIQ] */
    PeakEntry *t_e = (PeakEntry *)(nl_uniq + PYA_MAX_UNIQ); /* [cap + PYA_TABLE_PAD] */
    PrefixState *pre = (PrefixState *)(t_e + cap + PYA_TABLE_PAD);   /* [2][64], only if `prefix` */

    if (b.status[psm] != PYA_ST_OK) return;
    const uint32_t N = b.n_sig[psm];
    if (N == 0) return;

    /* stage the retained-peak table */
    PeakTable tab;
    stage_peak_table(b, psm, t_e, &tab);
    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = cfg->n_nl;
    env.nl_present = nl_present;
    env.nl_uniq = nl_uniq;
    if (env.n_nl) {
        for (int i = lane; i < 256; i += 64) nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) nl_uniq[lane] = cfg->uniq[lane];
    }
    wave_lds_sync();
    grid_build(&tab, grid);

    const Residues res = load_residues(b, cfg, psm);
    env.L = res.L;
    env.zmax = b.max_charge[psm];
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];
    const bool both_dirs = cfg->n_fwd > 0 && cfg->n_fwd < cfg->n_types;
    wave_lds_sync();
    /* localize looks up its few ions in the global table: leave it the grid (one 512-byte store) */
    ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)grid)[lane];

    int lut_fail = 0;
    const bool split = N <= 32 && both_dirs;     /* lanes 0..31 forward, 32..63 backward */
    const bool simple = walk_is_simple(env);
    const int n_sites = __popcll(res.site_mask);
    const bool shared = PREFIX && N >= 128 && n_sites >= PREFIX_SITES + 2;
    int stop[2] = {0, 0};
    if (shared) {
        /* steps [0, stop) of a direction cover exactly its first PREFIX_SITES sites */
        stop[0] = nth_set_bit(res.site_mask, PREFIX_SITES);
        stop[1] = res.L - 1 - nth_set_bit(res.site_mask, n_sites - 1 - PREFIX_SITES);
        for (int dir = 0; dir < 2; dir++) {
            if (dir == 0 ? cfg->n_fwd == 0 : cfg->n_fwd == cfg->n_types) continue;
            if (stop[dir] > res.L - 1) stop[dir] = res.L - 1;
            const uint64_t pbits = dir == 0 ? (uint64_t)lane : (__brevll((uint64_t)lane) >> (64 - n_sites));
            const uint64_t pmask = deposit_sites(pbits, res.site_mask);
            WalkState st = {0.f, 0u};
            Hist h = {0ull, 0ull, 0ull};
            uint32_t nf = 0;
            if (simple) walk_simple_range(env, res, tab, pmask, dir, true, 0, stop[dir], st, h, nf);
            else walk_range(env, res, tab, pmask, dir, true, 0, stop[dir], st, h, nf);
            PrefixState ps;
            ps.running = st.running;
            ps.nl_state = st.nl_state;
            ps.nfrag = nf;
            ps.pad = 0;
            ps.ha = h.a;
            ps.hb = h.b;
            ps.hc = h.c;
            pre[dir * 64 + lane] = ps;
        }
        wave_lds_sync();
    }
    for (uint32_t sbase = 0; sbase < N; sbase += 64) {
        const uint32_t s = split ? (uint32_t)(lane & 31) : sbase + lane;
        const bool active = s < N;
        const uint64_t bits = active ? order[s] : 0ull;
        const uint64_t resmask = deposit_sites(bits, res.site_mask);
        Hist h = {0ull, 0ull, 0ull};
        uint32_t nfrag = 0;
        if (shared) {
            for (int dir = 0; dir < 2; dir++) {
                if (dir == 0 ? cfg->n_fwd == 0 : cfg->n_fwd == cfg->n_types) continue;
                const uint32_t pat = dir == 0 ? (uint32_t)(bits & 63ull)
                                              : (uint32_t)((__brevll(bits) >> (64 - n_sites)) & 63ull);
                const PrefixState ps = pre[dir * 64 + pat];
                WalkState st = {ps.running, ps.nl_state};
                h.a += ps.ha;
                h.b += ps.hb;
                h.c += ps.hc;
                nfrag += ps.nfrag;
                if (simple) walk_simple_range(env, res, tab, resmask, dir, active, stop[dir], res.L - 1, st, h, nfrag);
                else walk_range(env, res, tab, resmask, dir, active, stop[dir], res.L - 1, st, h, nfrag);
            }
        } else if (simple) {
            if (split) {
                walk_simple(env, res, tab, resmask, lane >> 5, active, h, nfrag);
                fold_upper_half(h, nfrag);
            } else {
                if (cfg->n_fwd > 0) walk_simple(env, res, tab, resmask, 0, active, h, nfrag);
                if (cfg->n_fwd < cfg->n_types) walk_simple(env, res, tab, resmask, 1, active, h, nfrag);
            }
        } else if (split) {
            walk(env, res, tab, resmask, lane >> 5, active, h, nfrag);
            fold_upper_half(h, nfrag);
        } else {
            if (cfg->n_fwd > 0) walk(env, res, tab, resmask, 0, active, h, nfrag);
            if (cfg->n_fwd < cfg->n_types) walk(env, res, tab, resmask, 1, active, h, nfrag);
        }

        if (active && (!split || lane < 32)) {
            /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
            uint32_t cum[PYA_NTOP];
            uint32_t acc = 0;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                acc += hist_get(h, d);
                cum[d] = acc;
            }
            float ws = -1.f;
            if (nfrag <= b.lut_n_max) {
                double sum = 0.;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) {
                    float sc = lut_score(b, (uint32_t)d, cum[d], nfrag);
                    float prod = cfg->weights[d] * sc;                    /* float product ...   */
                    sum = sum + (double)prod;                             /* ... double sum      */
                }
                ws = (float)sum;
            } else {
                lut_fail = 1;
            }
            b.ws[s0 + s] = ws;
            if (b.rec) {
                uint32_t *rec = b.rec + (s0 + s) * PYA_REC_WORDS;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) rec[d >> 1] = cum[d] | (cum[d + 1] << 16);
                rec[5] = nfrag;
            }
        }
    }
    if (__any(lut_fail) && lane == 0) b.status[psm] = PYA_ST_LUT_RANGE;
}

extern "C" size_t pya_score_lds_bytes(uint32_t cap, uint32_t prefix) {
    return ((size_t)cap + PYA_TABLE_PAD) * 8 + 512 + PYA_GRID_CELLS * 2 + PYA_MAX_UNIQ * 4 + 64 +
           (prefix ? 2 * 64 * sizeof(PrefixState) : 0);
}

extern "C" int pya_launch_score(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                                uint32_t prefix, hipStream_t stream) {
    if (n_ids == 0) return 0;
    /* spectra near the 8192-peak limit need more than the default 64 KB of dynamic LDS */
    hipError_t e = prefix ? hipFuncSetAttribute((const void *)pya_score_signatures_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)pya_score_lds_bytes(cap, 1))
                          : hipFuncSetAttribute((const void *)pya_score_signatures_kernel<false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                                (int)pya_score_lds_bytes(cap, 0));
    if (e != hipSuccess) return (int)e;
    if (prefix)
        hipLaunchKernelGGL(pya_score_signatures_kernel<true>, dim3(n_ids), dim3(64), pya_score_lds_bytes(cap, 1),
                           stream, *b, d_ids, n_ids, cap);
    else
        hipLaunchKernelGGL(pya_score_signatures_kernel<false>, dim3(n_ids), dim3(64), pya_score_lds_bytes(cap, 0),
                           stream, *b, d_ids, n_ids, cap);
    return (int)hipGetLastError();
}
