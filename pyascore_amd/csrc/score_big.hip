/* score_big.hip -- score_signatures for PSMs with more than a thousand site assignments and the
 * plain scorer settings (no neutral losses, fragment charge 1, ion types of both directions, one
 * each): one PSM per 8-wavefront workgroup, fragment tree shared two levels deep.
 *
 * With C(n,k) in the thousands almost every fragment is shared by many signatures: a fragment's
 * m/z depends only on the pattern of the modifiable residues it contains, and the reference walks
 * the signatures as a tree for that reason (cpp/Ascore.cpp:69-109, cpp/ModifiedPeptide.cpp:458-471).
 * score_signatures shares the first 6 sites of each direction (64 patterns, one wavefront).  Here
 * the first TEN sites are shared, in two levels: level 1 = the 64 patterns of sites 0..5, level 2 =
 * the up to 1024 patterns of sites 0..9, each resumed from its level-1 parent; a signature resumes
 * from its level-2 pattern and walks only the rest.  Every state is the same sequence of float32
 * additions as a walk from the start, so the results are bit-identical.  For 30-mers with 5 of 15
 * sites (3003 signatures x 2 directions x 29 steps) that is 1.2 k + 10 k + 54 k lookups instead of
 * 1.5 k + 102 k.  The level-2 table (2 x 1024 x 16 B) is what a single wavefront cannot afford (LDS
 * decides this kernel's occupancy), hence the workgroup: eight wavefronts share one peak table,
 * one grid and both tables, and split the patterns and the signatures between them.
 *
 * Output as score_signatures: ws, count records, grid, the summary of the scores.
 */
#include "score_core.hip.h"

#define BIG_WAVES 8
#define BIG_SITES1 6
#define BIG_SITES2 10

/* (the level-2 table is indexed by the 10-site pattern itself; a dense table -- 638 instead of 1024
 * entries per direction for k = 5, 4 instead of 3 workgroups per CU -- was measured and lost to its
 * index arithmetic: 6.05 vs 5.66 ms on 50 000 PSMs of 3003 signatures) */
static inline size_t score_big_lds_bytes(uint32_t cap, uint32_t pos_cap) {
    return PYA_GRID_CELLS * 2 + 64 * 8 + ((size_t)cap + PYA_TABLE_PAD) * 8 + 2 * 64 * sizeof(PrefixCompact) +
           2 * 1024 * sizeof(PrefixCompact) + (size_t)BIG_WAVES * (PYA_NTOP / 2 * 64 * 4) + BIG_WAVES * 16 +
           (size_t)PYA_NTOP * (2 * pos_cap + 1) * 4 + 64;             /* + the score-table row of the PSM */
}

/* rank counts of column `lane` of a wave's histogram, packed in 8-bit fields (<= 63 per direction) */
DEV void pack_counts(const uint32_t *cnt, int lane, uint64_t *lo, uint32_t *hi) {
    uint64_t l = 0ull;
#pragma unroll
    for (int d = 0; d < 8; d++) l |= (uint64_t)hist_count(cnt, lane, d) << (d * 8);
    *lo = l;
    *hi = hist_count(cnt, lane, 8) | (hist_count(cnt, lane, 9) << 8);
}

__global__ __launch_bounds__(64 * BIG_WAVES, 6) void pya_score_big_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids,
                                                                       uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    const DevConfig *cfg = b.cfg;

    uint16_t *grid = (uint16_t *)lds_raw;                        /* [PYA_GRID_CELLS] */
    float2 *resd = (float2 *)(lds_raw + PYA_GRID_CELLS * 2);     /* [64] */
    PeakEntry *t_e = (PeakEntry *)(lds_raw + PYA_GRID_CELLS * 2 + 64 * 8);
    unsigned char *tail = lds_raw + PYA_GRID_CELLS * 2 + 64 * 8 + ((size_t)cap + PYA_TABLE_PAD) * 8;
    PrefixCompact *l1 = (PrefixCompact *)tail;                   /* [2][64]   */
    PrefixCompact *l2 = l1 + 2 * 64;                             /* [2][1024] indexed by the 10-site pattern */
    uint32_t *cnt_all = (uint32_t *)(l2 + 2 * 1024);             /* [BIG_WAVES][5][64] */
    uint32_t *tops = cnt_all + BIG_WAVES * (PYA_NTOP / 2 * 64);  /* [BIG_WAVES][4] */
    uint32_t *cnt = cnt_all + wave * (PYA_NTOP / 2 * 64);
    float *lutl = (float *)(tops + BIG_WAVES * 4);               /* [10][nfrag + 1] */

    if (b.status[psm] != PYA_ST_OK) return;                      /* (uniform over the workgroup) */
    const uint32_t N = b.n_sig[psm];
    if (N == 0) return;
    /* every wavefront reads the peptide for itself (registers: site mask, length); wavefront 0 stages
     * what is shared */
    const Residues res = load_residues(b, cfg, psm);
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];
    const int L = res.L;
    PeakTable tab;
    {
        const int64_t p0 = b.ret_off[psm];
        const int R = (int)b.ret_n[psm];
        copy_peak_table(b.ret + p0, R, t_e, tid, 64 * BIG_WAVES);
        tab.e = t_e;
        tab.g_cell = nullptr;
        tab.g_e = b.ret + p0;
        tab.n = R;
        tab.err = cfg->mz_error;
        tab.half_check = false;                                  /* (the host sends mz_error > 0.49 elsewhere) */
    }
    if (wave == 0) stage_residues(res, resd, nullptr);
    {
        /* the one row of the score table every signature of this PSM reads: 10 x (nfrag + 1) floats */
        const uint32_t nf = 2u * (uint32_t)(L - 1);
        if (nf <= b.lut_n_max) {
            const float *src = b.lut + lut_row(nf);
            for (uint32_t i = (uint32_t)tid; i < PYA_NTOP * (nf + 1); i += 64 * BIG_WAVES) lutl[i] = src[i];
        }
    }
    __syncthreads();
    /* the grid: cell geometry in every wavefront's registers, cells written by wavefront 0 */
    if (wave == 0) {
        grid_build(&tab, grid);
    } else {
        tab.cell = grid;
        if (tab.n > 0) grid_params(&tab, t_e[0].mz, t_e[tab.n - 1].mz);
        else { tab.base = 0.f; tab.inv_w = 0.f; tab.nb = 0.f; tab.last_cell = 0; }
    }
    __syncthreads();
    if (wave == 0) ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)grid)[lane];

    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = 0;
    env.nl_present = nullptr;
    env.nl_uniq = nullptr;
    env.resd = resd;
    env.resn = nullptr;
    env.cnt = cnt;
    env.L = L;
    env.zmax = 1;
    const int n_sites = __popcll(res.site_mask);
    /* steps [0, stop1) of a direction cover exactly its first 6 sites, [0, stop2) its first 10 */
    int stop1[2], stop2[2];
    stop1[0] = nth_set_bit(res.site_mask, BIG_SITES1);
    stop1[1] = L - 1 - nth_set_bit(res.site_mask, n_sites - 1 - BIG_SITES1);
    stop2[0] = nth_set_bit(res.site_mask, BIG_SITES2);
    stop2[1] = L - 1 - nth_set_bit(res.site_mask, n_sites - 1 - BIG_SITES2);
    for (int d = 0; d < 2; d++) {
        if (stop1[d] > L - 1) stop1[d] = L - 1;
        if (stop2[d] > L - 1) stop2[d] = L - 1;
    }
    const int k = b.n_of_mod[psm];

    /* ---- level 1: wavefront d walks the 64 patterns of the first 6 sites of direction d ---- */
    if (wave < 2) {
        const int dir = wave;
        const uint64_t pbits = dir == 0 ? (uint64_t)lane : (__brevll((uint64_t)lane) >> (64 - n_sites));
        WalkState st = {0.f, 0u};
        hist_clear(env);
        walk_simple_range(env, tab, deposit_sites(pbits, res.site_mask), dir, true, 0, stop1[dir], st);
        wave_lds_sync();
        PrefixCompact pc;
        pc.running = st.running;
        pack_counts(cnt, lane, &pc.lo, &pc.hi);
        l1[dir * 64 + lane] = pc;
    }
    __syncthreads();
    /* ---- level 2: the patterns of the first 10 sites that a signature can have (at most k modified,
     * enough sites left for the rest), resumed from level 1 ---- */
    for (int base = 0; base < 2 * 1024; base += 64 * BIG_WAVES) {
        const int item = base + tid;                             /* direction * 1024 + pattern */
        const int dir = item >> 10;                              /* (uniform within a wavefront: 64 | 1024) */
        const uint32_t c = (uint32_t)item & 1023u;
        const int m = __popc(c);
        const bool valid = m <= k && k - m <= n_sites - BIG_SITES2;
        if (!__any(valid)) continue;                             /* (no __syncthreads inside this loop) */
        const uint64_t pbits = dir == 0 ? (uint64_t)c : (__brevll((uint64_t)c) >> (64 - n_sites));
        const PrefixCompact par = l1[dir * 64 + (c & 63u)];
        WalkState st = {par.running, 0u};
        hist_clear(env);
        walk_simple_range(env, tab, deposit_sites(pbits, res.site_mask), dir, valid, stop1[dir], stop2[dir], st);
        wave_lds_sync();
        PrefixCompact pc;
        pc.running = st.running;
        pack_counts(cnt, lane, &pc.lo, &pc.hi);
        pc.lo += par.lo;                                         /* fields stay below 256: <= 63 per direction */
        pc.hi += par.hi;
        if (valid) l2[item] = pc;
        wave_lds_sync();
    }
    __syncthreads();
    /* ---- the signatures: resume from the level-2 patterns, walk the rest of both directions ---- */
    int lut_fail = 0;
    uint32_t top_u = 0, top_n = 0, top_i = 0xffffffffu;
    const uint32_t nfrag = 2u * (uint32_t)(L - 1);
    for (uint32_t sbase = 0; sbase < N; sbase += 64 * BIG_WAVES) {
        const uint32_t s = sbase + (uint32_t)tid;
        const bool active = s < N;
        const uint64_t bits = active ? order[s] : 0ull;
        const uint64_t resmask = deposit_sites(bits, res.site_mask);
        const PrefixCompact p0 = l2[(uint32_t)(bits & 1023ull)];
        const PrefixCompact p1 = l2[1024u + (uint32_t)((__brevll(bits) >> (64 - n_sites)) & 1023ull)];
        WalkState st0 = {p0.running, 0u}, st1 = {p1.running, 0u};
        const uint64_t p8lo = p0.lo + p1.lo;
        const uint32_t p8hi = p0.hi + p1.hi;
        hist_clear(env);
        walk_simple_both(env, tab, resmask, active, stop2[0], L - 1, st0, stop2[1], L - 1, st1);
        wave_lds_sync();
        if (active) {
            /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
            uint32_t cum[PYA_NTOP];
            uint32_t acc = 0;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                acc += hist_count(cnt, lane, d) + (d < 8 ? (uint32_t)(p8lo >> (d * 8)) & 0xffu : (p8hi >> ((d - 8) * 8)) & 0xffu);
                cum[d] = acc;
            }
            float ws = -1.f;
            if (nfrag <= b.lut_n_max) {
                double sum = 0.;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) {
                    const float sc = lutl[(uint32_t)d * (nfrag + 1) + cum[d]];
                    const float prod = cfg->weights[d] * sc;              /* float product ...   */
                    sum = sum + (double)prod;                             /* ... double sum      */
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
        wave_lds_sync();
    }
    /* ---- summary of the scores over the eight wavefronts ---- */
    {
        const uint32_t kmax = wave_max_u32(top_n ? top_u : 0u);
        const bool mine = top_n && top_u == kmax;
        const int n_max = wave_sum_i32(mine ? (int)top_n : 0);
        const uint32_t first = wave_min_u32(mine ? top_i : 0xffffffffu);
        if (lane == 0) {
            tops[wave * 4 + 0] = kmax;
            tops[wave * 4 + 1] = (uint32_t)n_max;
            tops[wave * 4 + 2] = first;
            tops[wave * 4 + 3] = (uint32_t)(__any(lut_fail) ? 1 : 0);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t kmax = 0, n_max = 0, first = 0xffffffffu, failed = 0;
        for (int wv = 0; wv < BIG_WAVES; wv++) {
            const uint32_t ku = tops[wv * 4], kn = tops[wv * 4 + 1], kf = tops[wv * 4 + 2];
            failed |= tops[wv * 4 + 3];
            if (kn == 0) continue;
            if (n_max == 0 || ku > kmax) {
                kmax = ku;
                n_max = kn;
                first = kf;
            } else if (ku == kmax) {
                n_max += kn;
                first = kf < first ? kf : first;
            }
        }
        uint32_t *t = b.ws_top + (size_t)psm * 4;
        t[0] = kmax;
        t[1] = n_max;
        t[2] = first;
        if (failed) b.status[psm] = PYA_ST_LUT_RANGE;
    }
}

extern "C" size_t pya_score_big_lds_bytes(uint32_t cap, uint32_t pos_cap) { return score_big_lds_bytes(cap, pos_cap); }

/* pos_cap: the largest L - 1 of the launch (sizes the score-table row kept in LDS) */
extern "C" int pya_launch_score_big(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                                    uint32_t pos_cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = score_big_lds_bytes(cap, pos_cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_score_big_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_score_big_kernel, dim3(n_ids), dim3(64 * BIG_WAVES), lds, stream, *b, d_ids, n_ids, cap);
    return (int)hipGetLastError();
}
