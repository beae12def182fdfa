/* score_cnt.hip -- score_signatures for PSMs with MANY site assignments under the plain scorer settings (no neutral
 * losses, fragment charge 1, one ion type per direction, both directions, mz_error <= 0.49): one PSM per wavefront,
 * every fragment decided by the count-node table of walk_core.hip.h instead of looked up by a walker.
 *
 * Replaces, for those PSMs, the FragmentGraph walk of Ascore::accumulateCounts and Ascore::calculateFullScores
 * (cpp/Ascore.cpp:53-139, cpp/ModifiedPeptide.cpp:126-150, :326-609) with
 *   1. the exact envelopes of the running sums per (direction, step, modified residues so far)       cnt_envelopes
 *   2. one peak lookup per such node: rank for every walker, or "marked"                             cnt_table_entry
 *   3. prefix sums of the nodes' count increments over the steps, taken at the sites' steps          site_prefix_sums
 *   4. one 16-byte entry per (modification, site) that serves both directions                        the table G
 *   5. per site assignment: k reads of G, ten score-table reads, the weighted sum.
 * A PSM with C(12,4) = 495 site assignments of a 40-mer costs 2 x 5 x 39 = 390 node lookups and 495 x 4 table reads
 * where the walkers made 495 x 78 = 38 610 lookups (fewer with the shared prefixes of score_signatures, never fewer than
 * one per distinct tree node).  Results are those of score_signatures bit for bit: a node whose decision could depend on
 * the walker (a peak within the few ulps between the windows of the smallest and the largest running sum) is marked, and
 * a site assignment through a marked node is walked with its own sums (walk_cnt_both).
 *
 * Output as score_signatures: ws, count records, grid, the summary of the scores.
 */
#include "score_core.hip.h"

struct CntLds {
    uint16_t *grid;
    float2 *resd;
    PeakEntry *t_e;
    uint4 *cum_lut;
    uint8_t *T;          /* [2][pos_cap][kc] */
    uint8_t *site_pos;   /* [64] */
    float2 *env;         /* [2][k_cap + 1][pos_cap]          } the same bytes: the envelopes are dead */
    uint4 *psite;        /* [2][k_cap + 1][n_cap + 1]        } when the prefix sums are written      */
    uint4 *G;            /* [k_cap * n_cap + 1] */
};
__host__ __device__ static inline size_t score_cnt_work_bytes(uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap) {
    const size_t env = (size_t)2 * (k_cap + 1) * pos_cap * sizeof(float2);
    const size_t tabs = ((size_t)2 * (k_cap + 1) * (n_cap + 1) + (size_t)k_cap * n_cap + 1) * sizeof(uint4);
    return env > tabs ? env : tabs;
}
__host__ __device__ static inline size_t score_cnt_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap, uint32_t n_cap) {
    return PYA_GRID_CELLS * 2 + ((((size_t)pos_cap + 1) * 8 + 15) & ~(size_t)15) + ((size_t)cap + PYA_TABLE_PAD) * 8 + 16 * sizeof(uint4) +
           (((size_t)2 * pos_cap * kc + 15) & ~(size_t)15) + 64 + score_cnt_work_bytes(pos_cap, k_cap, n_cap) + 16;
}
DEV CntLds cnt_carve(unsigned char *raw, uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap, uint32_t n_cap) {
    CntLds c;
    c.grid = (uint16_t *)raw;
    size_t o = PYA_GRID_CELLS * 2;
    c.resd = (float2 *)(raw + o);
    o += (((size_t)pos_cap + 1) * 8 + 15) & ~(size_t)15;
    c.t_e = (PeakEntry *)(raw + o);
    o += ((size_t)cap + PYA_TABLE_PAD) * 8;
    c.cum_lut = (uint4 *)(raw + o);
    o += 16 * sizeof(uint4);
    c.T = raw + o;
    o += ((size_t)2 * pos_cap * kc + 15) & ~(size_t)15;
    c.site_pos = raw + o;
    o += 64;
    c.env = (float2 *)(raw + o);
    c.psite = (uint4 *)(raw + o);
    c.G = c.psite + (size_t)2 * (k_cap + 1) * (n_cap + 1);
    return c;
}

/* P(d, j, e) (walk_core.hip.h) at the steps the sites enter at, and the row totals: one row (d, j) at a time, one step per
 * lane, an inclusive scan, then lane i (< n_sites) picks the value below its site's step.  Words: x, y = depths 0-7,
 * z = depths 8-9 in its low half and the MARKED nodes in its high half (at most 63 either).
 * psite[(d * (k + 1) + j) * (n_sites + 1) + i] = P(d, j, min(step of site i in direction d, L - 1)), entry n_sites = P(d, j, L - 1). */
DEV void site_prefix_sums(const CntLds &c, uint32_t pos_cap, uint32_t kc, int L, int k, int n_sites) {
    const int lane = lane_id();
    const int Lm1 = L - 1;
    /* peptides of up to 33 residues (most): two rows at a time, one per half of the wavefront */
    const bool two = Lm1 <= 32 && n_sites <= 32;
    const int half = two ? lane >> 5 : 0, sl = two ? lane & 31 : lane, base = two ? (lane & 32) : 0;
    const int pos = sl < n_sites ? (int)c.site_pos[sl] : 0;
    const int rows = 2 * (k + 1);
    for (int r0 = 0; r0 < rows; r0 += two ? 2 : 1) {
        const int row = r0 + half;
        const bool row_on = row < rows;
        const int d = row_on ? row / (k + 1) : 0, j = row_on ? row - d * (k + 1) : 0;
        uint32_t x = 0, y = 0, z = 0;
        if (sl < Lm1 && row_on) {
            const uint32_t ent = c.T[((size_t)d * pos_cap + sl) * kc + j];
            const uint4 inc = c.cum_lut[ent & 15u];
            x = inc.x;
            y = inc.y;
            z = inc.z | ((ent >> 7) << 16);
        }
        if (two) {
            x = wave_incl_scan_u32<true>(x);
            y = wave_incl_scan_u32<true>(y);
            z = wave_incl_scan_u32<true>(z);
        } else {
            x = wave_incl_scan_u32<false>(x);
            y = wave_incl_scan_u32<false>(y);
            z = wave_incl_scan_u32<false>(z);
        }
        /* lane i of the row's half: the sum over the steps below e = min(step of site i, L - 1) = the inclusive value of step e - 1 */
        const int st = d ? Lm1 - pos : pos;
        const int e = st < Lm1 ? st : Lm1;
        const int src = base + (e > 0 ? e - 1 : 0), last = base + (Lm1 > 0 ? Lm1 - 1 : 0);
        uint32_t px = (uint32_t)__shfl((int)x, src, 64), py = (uint32_t)__shfl((int)y, src, 64), pz = (uint32_t)__shfl((int)z, src, 64);
        if (e == 0) px = py = pz = 0u;
        const uint32_t tx = (uint32_t)__shfl((int)x, last, 64), ty = (uint32_t)__shfl((int)y, last, 64), tz = (uint32_t)__shfl((int)z, last, 64);
        if (row_on) {
            uint4 *out = c.psite + (size_t)row * (n_sites + 1);
            if (sl < n_sites) out[sl] = make_uint4(px, py, pz, 0u);
            if (sl == 0) out[n_sites] = make_uint4(tx, ty, tz, 0u);
        }
    }
}

/* G(t, site) and the constant (walk_core.hip.h: cnt_site_table), from the prefix sums at the sites */
DEV void site_table(const CntLds &c, int k, int n_sites) {
    const int lane = lane_id();
    const int W = n_sites + 1;
    for (int i = lane; i <= k * n_sites; i += 64) {
        uint4 g;
        if (i == k * n_sites) {
            const uint4 a = c.psite[(size_t)k * W + n_sites], q = c.psite[(size_t)(k + 1 + k) * W + n_sites];
            g = make_uint4(a.x + q.x, a.y + q.y, a.z + q.z, 0u);
        } else {
            const int t = i / n_sites + 1, site = i - (t - 1) * n_sites, tb = k + 1 - t;
            const uint4 f0 = c.psite[(size_t)(t - 1) * W + site], f1 = c.psite[(size_t)t * W + site];
            const uint4 b0 = c.psite[(size_t)(k + 1 + tb - 1) * W + site], b1 = c.psite[(size_t)(k + 1 + tb) * W + site];
            g = make_uint4((f0.x - f1.x) + (b0.x - b1.x), (f0.y - f1.y) + (b0.y - b1.y), (f0.z - f1.z) + (b0.z - b1.z), 0u);
        }
        c.G[i] = g;
    }
}

DEV void score_cnt_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap,
                        uint32_t n_cap) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const CntLds c = cnt_carve(lds_raw, cap, pos_cap, kc, k_cap, n_cap);
    /* (r06: the prologue's loads in two rounds -- device_common.hip.h: load_desc) */
    const LetterRegs letters = load_letter_regs(cfg);
    const PsmDesc dsc = load_desc(b, psm);
    const int status0 = b.status[psm];
    const int R0 = (int)b.ret_n[psm];
    if (status0 != PYA_ST_OK) return;
    const uint32_t N = dsc.N;
    if (N == 0) return;
    const Residues res = load_residues_desc(b, cfg, dsc, letters);
    const uint64_t *order = b.order_tab + dsc.order_off;
    const int64_t s0 = dsc.sig0;
    const int L = res.L, k = dsc.k, n_sites = __popcll(res.site_mask);
    /* the work areas are carved for the launch's caps (host_plan.cpp: the bucket's k_max / ns_max): a PSM beyond them would
     * overrun the envelopes and the (t, site) table -- the host never lists one here; if a routing change ever does, the PSM
     * fails loudly instead (r05 advisor) */
    if ((uint32_t)k > k_cap || (uint32_t)n_sites > n_cap || (uint32_t)(L - 1) > pos_cap || (uint32_t)k + 1u > kc) {
        if (lane == 0) b.status[psm] = PYA_ST_ROUTE_CAPS;
        return;
    }
    PeakTable tab;
    stage_peak_table_at(b, dsc.ret0, R0, c.t_e, &tab);
    stage_residues(res, c.resd, nullptr);
    if (lane < 16) c.cum_lut[lane] = fused_cum_entry((uint32_t)lane);
    if ((res.site_mask >> lane) & 1ull) c.site_pos[mask_rank(res.site_mask)] = (uint8_t)lane;
    for (uint32_t i = lane; i < (uint32_t)((2 * pos_cap * kc + 15) & ~15u) / 4u; i += 64) ((uint32_t *)c.T)[i] = 0x0f0f0f0fu;
    wave_lds_sync();
    grid_build(&tab, c.grid);
    wave_lds_sync();
    ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)c.grid)[lane];

    /* 1, 2: envelopes, then one lookup per node */
    cnt_envelopes(res, k, pos_cap, c.env);
    wave_lds_sync();
    double A0 = 0., B0 = 0., A1 = 0., B1 = 0.;
    type_constants(cfg->types[0], &A0, &B0);
    type_constants(cfg->types[cfg->n_fwd], &A1, &B1);
    {
        const uint32_t per_dir = (uint32_t)(k + 1) * (uint32_t)(L - 1);
        const FastDiv divL = fastdiv_make((uint32_t)(L - 1 > 0 ? L - 1 : 1));
        for (uint32_t i = (uint32_t)lane; i < 2u * per_dir; i += 64) {
            const uint32_t d = i >= per_dir ? 1u : 0u, r = i - d * per_dir, j = fastdiv(r, divL), st = r - j * (uint32_t)(L - 1);
            const float2 lh = c.env[(size_t)(d * (uint32_t)(k + 1) + j) * pos_cap + st];
            uint32_t ent = cnt_table_entry(tab, lh.x, lh.y, d ? A1 : A0, d ? B1 : B0);
            if ((b.debug & 0x40000000u) && lh.x <= lh.y) ent |= CNT_MARK;
            c.T[((size_t)d * pos_cap + st) * kc + j] = (uint8_t)ent;
#ifdef PYA_STAMPS                                              /* diagnostic build: reachable nodes looked up / marked (slots 56, 57) */
            if (b.stamps && lh.x <= lh.y) {
                atomicAdd(&b.stamps[56], 1ull);
                if (ent & CNT_MARK) atomicAdd(&b.stamps[57], 1ull);
            }
#endif
        }
    }
    wave_lds_sync();
    /* 3, 4 */
    site_prefix_sums(c, pos_cap, kc, L, k, n_sites);
    wave_lds_sync();
    site_table(c, k, n_sites);
    wave_lds_sync();

    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = 0;
    env.nl_present = nullptr;
    env.nl_uniq = nullptr;
    env.resd = c.resd;
    env.resn = nullptr;
    env.cnt = nullptr;
    env.L = L;
    env.zmax = 1;
    /* 5: the site assignments */
    int lut_fail = 0;
    uint32_t top_u = 0, top_n = 0, top_i = 0xffffffffu;
    const uint32_t nfrag = 2u * (uint32_t)(L - 1);
    const uint4 *row0 = c.G;
    const uint4 gconst = c.G[k * n_sites];
    for (uint32_t sbase = 0; sbase < N; sbase += 64) {
        const uint32_t s = sbase + (uint32_t)lane;
        const bool active = s < N;
        const uint64_t bits = active ? order[s] : 0ull;
        uint32_t ax = gconst.x, ay = gconst.y, az = gconst.z;
        {
            uint32_t m = (uint32_t)bits;
            const uint4 *row = row0;
            for (int t = 0; t < k; t++, row += n_sites) {
                const int site = active ? __builtin_ctz(m) : 0;
                m &= m - 1u;
                const uint4 g = row[site];
                ax += g.x;
                ay += g.y;
                az += g.z;
            }
        }
        CumCounts cc = {ax, ay, az & 0xffffu};
        const bool marked = active && (az >> 16) != 0u;
#ifdef PYA_STAMPS                                              /* ... site assignments / those through a marked node (58, 59) */
        if (b.stamps && active) {
            atomicAdd(&b.stamps[58], 1ull);
            if (marked) atomicAdd(&b.stamps[59], 1ull);
        }
#endif
        if (__any(marked)) {                                 /* (rare: a peak within a few ulps of some window end) */
            float run0 = 0.f, run1 = 0.f;
            CumCounts cw = {0u, 0u, 0u};
            walk_cnt_both(env, tab, c.cum_lut, c.T, pos_cap, kc, deposit_sites(bits, res.site_mask), 0, L - 1, run0, 0u, 0, L - 1, run1, 0u, cw);
            if (marked) cc = cw;
        }
        if (active) {
            /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
            uint32_t cum[PYA_NTOP];
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) cum[d] = cc.at(d);
            float ws = -1.f;
            if (nfrag <= b.lut_n_max) {
                double sum = 0.;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) {
                    const float sc = lut_score(b, (uint32_t)d, cum[d], nfrag);
                    const float prod = cfg->weights[d] * sc;                  /* float product ...   */
                    sum = sum + (double)prod;                                 /* ... double sum      */
                }
                ws = (float)sum;
            } else {
                lut_fail = 1;
            }
            b.ws[s0 + s] = ws;
            const uint32_t u = __float_as_uint(ws);
            if (ws >= 0.f && (top_n == 0 || u > top_u)) {
                top_u = u;
                top_n = 1;
                top_i = s;
            } else if (ws >= 0.f && u == top_u) {
                top_n++;
            }
            if (b.rec) {
                uint32_t *rec = b.rec + (s0 + s) * PYA_REC_WORDS;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) rec[d >> 1] = cum[d] | (cum[d + 1] << 16);
                rec[5] = nfrag;
            }
        }
    }
    if (__any(lut_fail) && lane == 0) b.status[psm] = PYA_ST_LUT_RANGE;
    {
        const uint32_t kmax = wave_max_u32(top_n ? top_u : 0u);
        const bool mine = top_n && top_u == kmax;
        const int n_max = wave_sum_i32(mine ? (int)top_n : 0);
        const uint32_t first = wave_min_u32(mine ? top_i : 0xffffffffu);
        if (lane == 0) {
            uint32_t *t = b.ws_top + (size_t)psm * 4;
            t[0] = kmax;
            t[1] = (uint32_t)n_max;
            t[2] = first;
        }
    }
}

#ifndef SCORE_CNT_WAVES
#define SCORE_CNT_WAVES 6
#endif
__global__ __launch_bounds__(64, SCORE_CNT_WAVES) void pya_score_cnt_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap,
                                                                          uint32_t pos_cap, uint32_t kc, uint32_t k_cap, uint32_t n_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    score_cnt_body(b, psm_ids[xcd_slot(blockIdx.x, n_ids)], lds_raw, cap, pos_cap, kc, k_cap, n_cap);
}

extern "C" size_t pya_score_cnt_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap, uint32_t n_cap) {
    return score_cnt_lds_bytes(cap, pos_cap, kc, k_cap, n_cap);
}

/* kc: a power of two >= 8 and > k_cap; k_cap / n_cap: the most modifications / modifiable residues of the launch's PSMs */
extern "C" int pya_launch_score_cnt(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap, uint32_t kc,
                                    uint32_t k_cap, uint32_t n_cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = score_cnt_lds_bytes(cap, pos_cap, kc, k_cap, n_cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_score_cnt_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_score_cnt_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, pos_cap, kc, k_cap, n_cap);
    return (int)hipGetLastError();
}
