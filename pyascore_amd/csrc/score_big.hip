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
#define BIG_T (64 * BIG_WAVES)

/* The winner among tied best PepScores, named by the kernel itself (inline mode: summary results only, C(n,k) <=
 * pya_big_inline_max()).  Half of cfg5's PSMs tie for the best score, and which of the tied site assignments
 * the reference reports is whatever libstdc++'s introsort leaves at the front: its emulation on ONE wavefront
 * over 3003 scores (the lean localize kernel's second pass) took 1.13 of cfg5's 6.5 ms.  Here, once the last
 * signature is scored, the two prefix tables and the eight rank histograms are dead -- 44 KB of the workgroup's
 * LDS -- and become   key f32[N] | idx u16[N] | lq u16[N] | rq u16[N] | 2 KB of chunk masks and counts,
 * on which all eight wavefronts run the left spine of the sort (wg_spine_front) and leave the winner's index in
 * the score summary; the localize kernel then never sorts.  In this mode no count record is written either --
 * 24 bytes x 3003 signatures x 50 000 PSMs = 3.6 GB per step on cfg5, read back for a dozen signatures per PSM:
 * the localize kernel recounts the few signatures it looks at (localize_core.hip.h: loc_recount). */
#define BIG_INLINE_AUX 2304
#define BIG_SORT_AUX 1600          /* what the spine really uses: two stop masks and two counts per chunk, five words */
#define BIG_CAND_REC 128           /* candidate mode: PepScores at words [0, items] of the PSM's score area, count records from this word on */
#define BIG_CAND_FLAG 0xC0DE0001u  /* fourth word of the score summary: the area holds candidate records, not N PepScores */

/* (the level-2 table is indexed by the 10-site pattern itself; a dense table -- 638 instead of 1024
 * entries per direction for k = 5, 4 instead of 3 workgroups per CU -- was measured and lost to its
 * index arithmetic: 6.05 vs 5.66 ms on 50 000 PSMs of 3003 signatures) */
/* (r04: the walkers keep their rank counts in registers -- walk_core.hip.h: CumCounts -- so the eight histograms, 10 KB,
 * are gone: 40 KB per workgroup on cfg5's shape, a fourth workgroup per CU) */
/* layout: grid | residues [pos_cap + 1] | peak table | level-2 table | count table | summary words | a region that holds
 * the level-1 table and the list of valid level-2 patterns while the tables are built, then the score-table row of the
 * PSM (38.1 KB + 8 B per retained peak for a 30-mer: four workgroups per CU up to 358 retained peaks) */
__host__ __device__ static inline size_t score_big_resd_bytes(uint32_t pos_cap) { return (((size_t)pos_cap + 1) * 8 + 15) & ~(size_t)15; }
__host__ __device__ static inline size_t score_big_tail_bytes(uint32_t pos_cap) {
    const size_t build = 2 * 64 * sizeof(PrefixCompact) + 1024 * sizeof(uint16_t), row = (size_t)PYA_NTOP * (2 * pos_cap + 1) * 4;
    return (build > row ? build : row) + 64;
}
/* the count-node table (walk_core.hip.h): one byte per (direction, step, modified residues so far) */
__host__ __device__ static inline size_t score_big_cnt_bytes(uint32_t pos_cap, uint32_t kc) { return ((size_t)2 * pos_cap * kc + 15) & ~(size_t)15; }
static inline size_t score_big_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc) {
    return PYA_GRID_CELLS * 2 + score_big_resd_bytes(pos_cap) + ((size_t)cap + PYA_TABLE_PAD) * 8 +
           2 * 1024 * sizeof(PrefixCompact) + 16 * sizeof(uint4) + BIG_WAVES * 16 + score_big_cnt_bytes(pos_cap, kc) +
           score_big_tail_bytes(pos_cap);
}

/* a table entry's counts: the CUMULATIVE counts of depths 0..7 in lo, 8..9 in hi, a byte each (they add without
 * unpacking: at most 126 fragments per site assignment here) */
DEV CumCounts entry_counts(const PrefixCompact &p) {
    CumCounts c = {(uint32_t)p.lo, (uint32_t)(p.lo >> 32), p.hi};
    return c;
}
DEV PrefixCompact make_entry(float running, const CumCounts &c) {
    PrefixCompact p;
    p.running = running;
    p.lo = (uint64_t)c.a | ((uint64_t)c.b << 32);
    p.hi = c.c;
    return p;
}


/* Front of libstdc++'s std::sort (descending by PepScore) over key[0..N), by the whole workgroup: the left
 * spine of introsort's partition tree -- __unguarded_partition_pivot on [0, l) until l <= 16 -- exactly as
 * localize_core.hip.h's sort_partition<true> runs it on one wavefront (same pivot choice, same cursor stops,
 * same pairing, only the left part written), with the stops ranked by a scan over per-chunk ballots instead of
 * queues.  Returns the pre-sort index of the front element; *out_of_depth when the depth limit would call for
 * the heap sort (the caller hands the PSM over). */
struct BigSortLds {
    float *key;
    uint16_t *idx, *lq, *rq;
    unsigned long long *ml, *mr;       /* [64] stop masks per chunk of 64 positions */
    uint32_t *cl, *cr;                 /* [64] stops per chunk */
    uint32_t *misc;                    /* [8]: nL, nR, swaps, cut, front */
};
DEV uint32_t wg_spine_front(const BigSortLds &s, int N, uint32_t kmax, bool *out_of_depth) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int depth = 0;
    for (int t = N; t > 1; t >>= 1) depth++;
    depth *= 2;
    int l = N;
    *out_of_depth = false;
    if (tid < 2) s.misc[2 + tid] = 0u;
    __syncthreads();
    /* three barriers per partition: every thread works out the pivot for itself (the element swapped to the front
     * is only written once everybody has taken its stops), every wavefront scans the chunk counts for itself, and
     * every thread derives the cut for itself from the swap counter (two counters, used alternately) */
    for (int it = 0; l > 16; it++) {
        if (depth == 0) {
            *out_of_depth = true;
            return 0u;
        }
        depth--;
        /* __move_median_to_first(0, 1, mid, l - 1): the pick, and the array as it looks after the swap */
        const int mid = l / 2;
        int pick;
        {
            const float a = s.key[1], bq = s.key[mid], c = s.key[l - 1];
            if (a > bq) {
                if (bq > c) pick = mid;
                else if (a > c) pick = l - 1;
                else pick = 1;
            } else if (a > c) pick = 1;
            else if (bq > c) pick = l - 1;
            else pick = mid;
        }
        const float pv = s.key[pick], k_front = s.key[0];
        const uint16_t i_pick = s.idx[pick], i_front = s.idx[0];
        const int nchunks = (l + 63) >> 6;
        for (int c = wave; c < nchunks; c += BIG_WAVES) {
            const int p = c * 64 + lane;
            float k = s.key[p < l ? p : l - 1];
            k = p == pick ? k_front : k;                     /* (position 0 is no left stop and holds the pivot for the right cursor) */
            k = p == 0 ? pv : k;
            const unsigned long long ml = __ballot(p >= 1 && p < l && !(k > pv));
            const unsigned long long mr = __ballot(p < l && !(pv > k));
            if (lane == 0) {
                s.ml[c] = ml;
                s.mr[c] = mr;
                s.cl[c] = (uint32_t)__popcll(ml);
                s.cr[c] = (uint32_t)__popcll(mr);
            }
        }
        __syncthreads();
        if (tid == 0) {                                      /* the swap itself, and the other partition's counter */
            s.key[0] = pv;
            s.idx[0] = i_pick;
            s.key[pick] = k_front;
            s.idx[pick] = i_front;
            s.misc[2 + ((it + 1) & 1)] = 0u;
        }
        /* stops before each chunk (left cursor ascending, right descending): every wavefront for itself */
        int nL, nR;
        uint32_t my_ol, my_or;
        {
            const int vl = lane < nchunks ? (int)s.cl[lane] : 0;
            my_ol = (uint32_t)wave_excl_scan_i32(vl, &nL);
            const int rc = nchunks - 1 - lane;               /* lane j takes the j-th chunk from the right */
            const int vr = rc >= 0 ? (int)s.cr[rc] : 0;
            my_or = (uint32_t)wave_excl_scan_i32(vr, &nR);
        }
        for (int c = wave; c < nchunks; c += BIG_WAVES) {
            const unsigned long long ml = s.ml[c], mr = s.mr[c];
            const uint32_t ol = (uint32_t)__builtin_amdgcn_readlane((int)my_ol, c);
            const uint32_t orr = (uint32_t)__builtin_amdgcn_readlane((int)my_or, nchunks - 1 - c);
            const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : (~0ull << (lane + 1));
            if ((ml >> lane) & 1ull) s.lq[ol + (uint32_t)__popcll(ml & below)] = (uint16_t)(c * 64 + lane);
            if ((mr >> lane) & 1ull) s.rq[orr + (uint32_t)__popcll(mr & above)] = (uint16_t)(c * 64 + lane);
        }
        __syncthreads();
        const int np = nL < nR ? nL : nR;
        /* the r-th stops are exchanged while the cursors have not met; left stops ascend and right stops descend,
         * so the exchanged pairs are r < m_sw, their left and right positions are disjoint sets, and only the left
         * element is written (the discarded right part keeps stale copies, as in the one-wavefront form) */
        int mine = 0;
        for (int r = tid; r < np; r += BIG_T) {
            const int a = s.lq[r], bpos = s.rq[r];
            if (a < bpos) {
                s.key[a] = s.key[bpos];
                s.idx[a] = s.idx[bpos];
                mine++;
            }
        }
        mine = wave_sum_i32(mine);
        if (lane == 0 && mine) atomicAdd(&s.misc[2 + (it & 1)], (uint32_t)mine);
        __syncthreads();
        const int m_sw = (int)s.misc[2 + (it & 1)];
        const int cand_l = m_sw < nL ? (int)s.lq[m_sw] : 0x7fffffff;
        const int cand_r = m_sw >= 1 ? (int)s.rq[m_sw - 1] : l;
        l = cand_l < cand_r ? cand_l : cand_r;
    }
    __syncthreads();
    /* front of the sorted list = left-most maximum of the left-most run */
    if (wave == 0) {
        uint32_t pos = 0xffffffffu;
        if (lane < l && __float_as_uint(s.key[lane]) == kmax) pos = (uint32_t)lane;
        pos = wave_min_u32(pos);
        if (lane == 0) s.misc[4] = s.idx[pos];
    }
    __syncthreads();
    return s.misc[4];
}

__host__ __device__ static inline uint32_t pya_big_inline_max_dev() {
    const size_t dead = 2 * 1024 * sizeof(PrefixCompact) + 16 * sizeof(uint4);     /* level-2 table | count table */
    const size_t n = (dead - BIG_INLINE_AUX) / 10;
    return (uint32_t)(n < 4096 ? n : 4096);                  /* (chunk masks and counts: 64 chunks of 64 positions) */
}
extern "C" uint32_t pya_big_inline_max(void) {
    /* key (4) + idx, lq, rq (3 x 2) bytes per signature in what the prefix tables and histograms leave */
    return pya_big_inline_max_dev();
}

/* the one row of the score table every site assignment of a PSM reads: 10 x (nfrag + 1) floats into LDS, TRANSPOSED --
 * [count][depth] -- so that a site assignment's ten reads share ONE base register and differ in the instruction's offset
 * field: with [depth][count] the compiler kept ten row bases in scalar registers, spilled them into a vector register's
 * lanes and paid a v_readlane + wait per read.  Threads t = 0 .. nt - 1 of the caller's choice. */
DEV void stage_score_row(const BatchDev &b, float *lutl, int L, int t, int nt) {
    const uint32_t nf = 2u * (uint32_t)(L - 1);
    if (nf > b.lut_n_max) return;
    const float *src = b.lut + lut_row(nf);
    const FastDiv divR = fastdiv_make(nf + 1u);
    for (uint32_t i = (uint32_t)t; i < PYA_NTOP * (nf + 1); i += (uint32_t)nt) {
        const uint32_t d = fastdiv(i, divR), c = i - d * (nf + 1u);
        lutl[c * PYA_NTOP + d] = src[i];
    }
}

struct BigLoc {
    uint32_t inline_on;              /* summary results only: no count records, ties for the best score resolved here */
};

/* one PSM, one workgroup (uniform control flow: it contains workgroup barriers) */
DEV void big_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t pos_cap, uint32_t kc, const BigLoc &loc) {
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int tid = (int)threadIdx.x;
    const DevConfig *cfg = b.cfg;

    uint16_t *grid = (uint16_t *)lds_raw;                        /* [PYA_GRID_CELLS] */
    float2 *resd = (float2 *)(lds_raw + PYA_GRID_CELLS * 2);     /* [pos_cap + 1] */
    PeakEntry *t_e = (PeakEntry *)(lds_raw + PYA_GRID_CELLS * 2 + score_big_resd_bytes(pos_cap));
    unsigned char *tail = (unsigned char *)t_e + ((size_t)cap + PYA_TABLE_PAD) * 8;
    PrefixCompact *l2 = (PrefixCompact *)tail;                   /* [2][1024] indexed by the 10-site pattern */
    uint4 *cum_lut = (uint4 *)(l2 + 2 * 1024);                   /* [16] rank -> increments of the cumulative counts */
    uint32_t *tops = (uint32_t *)(cum_lut + 16);                 /* [BIG_WAVES][4] */
    uint8_t *cnt_t = (uint8_t *)(tops + BIG_WAVES * 4);          /* [2][pos_cap][kc] the count-node table */
    float *lutl = (float *)(cnt_t + score_big_cnt_bytes(pos_cap, kc));   /* [nfrag + 1][10], once the tables are built; until then: */
    PrefixCompact *l1 = (PrefixCompact *)lutl;                   /* [2][64]   */
    uint16_t *vlist = (uint16_t *)(l1 + 2 * 64);                 /* [1024] the level-2 patterns a signature can have */

    STAMP_BEGIN();
    /* (r06: the prologue's loads in two rounds -- device_common.hip.h: load_desc) */
    const LetterRegs letters = load_letter_regs(cfg);
    const PsmDesc dsc = load_desc(b, psm);
    const int status0 = b.status[psm];
    const int R0 = (int)b.ret_n[psm];
    if (status0 != PYA_ST_OK) return;                            /* (uniform over the workgroup) */
    const uint32_t N = dsc.N;
    if (N == 0) return;
    const bool inl = loc.inline_on && !b.keep && N <= pya_big_inline_max_dev();
    /* every wavefront reads the peptide for itself (registers: site mask, length); wavefront 0 stages
     * what is shared */
    const Residues res = load_residues_desc(b, cfg, dsc, letters);
    const uint64_t *order = b.order_tab + dsc.order_off;
    const int64_t s0 = dsc.sig0;
    const int L = res.L;
    /* the count-node table (walk_core.hip.h) -- its envelopes need the residues only: wavefront 1 runs the recurrence (L - 1
     * dependent steps) while the others stage the peak table and wavefront 0 builds the grid */
    const int k = dsc.k;
    const int n_sites = __popcll(res.site_mask);
    uint4 *cntG = (uint4 *)l2;                                  /* [k * n_sites + 1] the per-site table (first: it outlives the prefix sums) */
    uint4 *cntP = cntG + (size_t)k * n_sites + 1;               /* [2][k + 1][L] prefix sums */
    float2 *envl = (float2 *)((unsigned char *)l2 + 16384);     /* [2][k + 1][pos_cap] */
    const bool use_cnt = !(b.debug & 0x8000u) && (uint32_t)k + 1u <= kc && k + 1 <= 31 && n_sites <= 32 &&
                         ((size_t)2 * (k + 1) * L + (size_t)k * n_sites + 1) * sizeof(uint4) <= 16384 &&
                         (size_t)2 * (k + 1) * pos_cap * sizeof(float2) <= 16384;
    /* r06 -- candidate mode (inline + count nodes): the PepScores never leave the workgroup.  They are kept at the END of the
     * level-2 table's room (dead on this route), the sort's queues lie between the per-site table and them, its chunk masks in
     * what the score-table row leaves of the tail -- so that the per-site table, the count nodes and the row all survive the
     * sort -- and once the winner is named, wavefront 0 scores its k (n - k) single-move competitors from the table again and
     * leaves their PepScores and count records (and the winner's) where the PepScores used to go: the finishing kernel reads
     * those and neither ranks combinations, gathers scores nor recounts (Ascore.cpp:212-254 needs nothing else). */
    const uint32_t N4 = (N + 3u) & ~3u, N8 = (N + 7u) & ~7u;
    const int n_free = n_sites - k, items = k * n_free;
    const size_t g_bytes = (((size_t)k * n_sites + 1) * sizeof(uint4) + 15) & ~(size_t)15;
    const size_t row_bytes = ((size_t)PYA_NTOP * (2u * (uint32_t)(L - 1) + 1u) * 4 + 15) & ~(size_t)15;
    const bool aux_in_tail = row_bytes + BIG_SORT_AUX <= score_big_tail_bytes(pos_cap) - 64;
    const bool cand = inl && use_cnt && items >= 1 && items <= 126 && (uint32_t)(BIG_CAND_REC + (items + 1) * PYA_REC_WORDS) <= N &&
                      g_bytes + (size_t)6 * N8 + (aux_in_tail ? 0 : BIG_SORT_AUX) + (size_t)4 * N4 <= 2 * 1024 * sizeof(PrefixCompact) &&
                      !(b.debug & 0x10000000u);
    float *wsl = (float *)((unsigned char *)l2 + 2 * 1024 * sizeof(PrefixCompact) - (size_t)4 * N4);
    if (use_cnt && wave == 1) cnt_envelopes(res, k, pos_cap, envl);
    PeakTable tab;
    {
        const int64_t p0 = dsc.ret0;
        const int R = R0;
        copy_peak_table(b.ret + p0, R, t_e, tid, 64 * BIG_WAVES);
        tab.e = t_e;
        tab.g_cell = nullptr;
        tab.g_e = b.ret + p0;
        tab.n = R;
        tab.err = cfg->mz_error;
        tab.half_check = false;                                  /* (the host sends mz_error > 0.49 elsewhere) */
    }
    if (wave == 0) stage_residues(res, resd, nullptr);
    if (tid < 16) cum_lut[tid] = fused_cum_entry((uint32_t)tid);
    if (use_cnt) {
        /* r06: what does not depend on the envelopes is done while wavefront 1 runs their recurrence (L - 1 dependent steps: the
         * others waited for it at the barrier below, then filled the table and staged the score row behind two more barriers) */
        for (uint32_t i = (uint32_t)tid; i < (uint32_t)score_big_cnt_bytes(pos_cap, kc) / 4u; i += BIG_T) ((uint32_t *)cnt_t)[i] = 0x0f0f0f0fu;
        if (wave >= 2) stage_score_row(b, lutl, L, tid - 128, BIG_T - 128);
    }
    __syncthreads();
    STAMP_T(b, 13, );
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
    STAMP_T(b, 14, );

    /* ---- the count-node table (walk_core.hip.h): one lookup per (direction, step, modified residues so far), then the
     * prefix sums over the steps.  The envelopes and the prefix sums go through the level-2 table's LDS, which this route
     * does not use (it needs neither prefix level): P at l2, the envelopes 16 KB behind it. ---- */
    if (use_cnt) {
        STAMP_T(b, 15, );
        double A0 = 0., B0 = 0., A1 = 0., B1 = 0.;
        type_constants(cfg->types[0], &A0, &B0);
        type_constants(cfg->types[cfg->n_fwd], &A1, &B1);
        const uint32_t per_dir = (uint32_t)(k + 1) * (uint32_t)(L - 1);
        for (uint32_t i = (uint32_t)tid; i < 2u * per_dir; i += BIG_T) {
            const uint32_t d = i >= per_dir ? 1u : 0u, r = i - d * per_dir, j = r / (uint32_t)(L - 1), st = r - j * (uint32_t)(L - 1);
            const float2 lh = envl[(size_t)(d * (uint32_t)(k + 1) + j) * pos_cap + st];
            uint32_t ent = cnt_table_entry(tab, lh.x, lh.y, d ? A1 : A0, d ? B1 : B0);
            if ((b.debug & 0x40000000u) && lh.x <= lh.y) ent |= CNT_MARK;      /* (every walker looks every fragment up itself: must agree) */
            cnt_t[((size_t)d * pos_cap + st) * kc + j] = (uint8_t)ent;
#ifdef PYA_STAMPS                                              /* diagnostic build: nodes looked up / marked (slots 50, 51) */
            if (b.stamps) {
                atomicAdd(&b.stamps[50], 1ull);
                if (ent & CNT_MARK) atomicAdd(&b.stamps[51], 1ull);
            }
#endif
        }
        __syncthreads();
        STAMP_T(b, 16, );
        cnt_prefix_sums(cnt_t, cum_lut, pos_cap, kc, L, k, cntP, wave, BIG_WAVES);
        if (wave == 1 && ((res.site_mask >> lane) & 1ull))       /* residue of the j-th modifiable one (first bytes of the envelopes' room) */
            ((uint8_t *)envl)[mask_rank(res.site_mask)] = (uint8_t)lane;
        __syncthreads();
        cnt_site_table(cntP, (const uint8_t *)envl, L, k, n_sites, cntG, tid, BIG_T);
        __syncthreads();
    }

    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = 0;
    env.nl_present = nullptr;
    env.nl_uniq = nullptr;
    env.resd = resd;
    env.resn = nullptr;
    env.cnt = nullptr;                                           /* (no histogram: register counts) */
    env.L = L;
    env.zmax = 1;
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

    /* ---- level 1: wavefront d walks the 64 patterns of the first 6 sites of direction d ---- */
    if (wave < 2 && !use_cnt) {
        const int dir = wave;
        const uint64_t pbits = dir == 0 ? (uint64_t)lane : (__brevll((uint64_t)lane) >> (64 - n_sites));
        float run = 0.f;
        CumCounts cum = {0u, 0u, 0u};
        walk_cum_range(env, tab, cum_lut, deposit_sites(pbits, res.site_mask), dir, 0, stop1[dir], run, cum);
        l1[dir * 64 + lane] = make_entry(run, cum);
    }
    if (!use_cnt) __syncthreads();                               /* (use_cnt is uniform over the workgroup) */
    STAMP_T(b, 9, );
    /* ---- level 2: the patterns of the first 10 sites that a signature can have (at most k modified,
     * enough sites left for the rest), resumed from level 1.  Which patterns those are depends on their number of
     * modified sites only (638 of the 1024 for 5 of 15): they are listed first -- in the LDS the score-table row
     * will take afterwards -- so that the wavefronts walk full rounds of them, not rounds with a third of the lanes
     * idle. ---- */
    const bool listed = true;
    uint32_t nv = 1024u;
    if (!use_cnt) {
        uint32_t before = 0;                                     /* valid patterns below this thread's, round by round */
        for (int r = 0; r < 1024 / BIG_T; r++) {
            const uint32_t c = (uint32_t)(r * BIG_T + tid);
            const int m = __popc(c);
            const bool valid = m <= k && k - m <= n_sites - BIG_SITES2;
            const uint64_t vm = __ballot(valid);
            if (lane == 0) tops[wave] = (uint32_t)__popcll(vm);
            __syncthreads();
            uint32_t off = before, tot = 0;
            for (int wv = 0; wv < BIG_WAVES; wv++) {
                const uint32_t n_w = tops[wv];
                off += wv < wave ? n_w : 0u;
                tot += n_w;
            }
            if (valid) vlist[off + (uint32_t)mask_rank(vm)] = (uint16_t)c;
            before += tot;
            __syncthreads();
        }
        nv = before;
    }
    const uint32_t nv64 = (nv + 63u) & ~63u;                     /* a wavefront's 64 items share their direction */
    for (uint32_t base = 0; !use_cnt && base < 2u * nv64; base += 64 * BIG_WAVES) {
        const uint32_t it2 = base + (uint32_t)tid;
        const int dir = it2 >= nv64 ? 1 : 0;                     /* (uniform within a wavefront) */
        const uint32_t q = it2 - (dir ? nv64 : 0u);
        const bool listed_on = q < nv && it2 < 2u * nv64;
        const uint32_t c = listed ? (listed_on ? (uint32_t)vlist[q] : 0u) : q;
        const int item = dir * 1024 + (int)c;                    /* direction * 1024 + pattern */
        const int m = __popc(c);
        const bool valid = listed_on && m <= k && k - m <= n_sites - BIG_SITES2;
        if (!__any(valid)) continue;                             /* (no __syncthreads inside this loop) */
        const uint64_t pbits = dir == 0 ? (uint64_t)c : (__brevll((uint64_t)c) >> (64 - n_sites));
        const PrefixCompact par = l1[dir * 64 + (c & 63u)];
        float run = par.running;
        CumCounts cum = entry_counts(par);                       /* (the walk adds to the parent's counts) */
        walk_cum_range(env, tab, cum_lut, deposit_sites(pbits, res.site_mask), dir, stop1[dir], stop2[dir], run, cum);
        if (valid) l2[item] = make_entry(run, cum);
    }
    if (!use_cnt) {
        __syncthreads();                                         /* (the list of patterns is dead: the row takes its place) */
        stage_score_row(b, lutl, L, tid, BIG_T);
        __syncthreads();
    }
    STAMP_T(b, 10, );
    /* ---- the signatures: resume from the level-2 patterns, walk the rest of both directions ---- */
    int lut_fail = 0;
    uint32_t top_u = 0, top_n = 0, top_i = 0xffffffffu, top_b = 0;     /* (top_b: the low word of that site assignment) */
    const uint32_t nfrag = 2u * (uint32_t)(L - 1);
    uint64_t bits_next = (uint32_t)tid < N ? order[tid] : 0ull;  /* (one round ahead: the load of a round travels during the round before) */
    for (uint32_t sbase = 0; sbase < N; sbase += 64 * BIG_WAVES) {
        const uint32_t s = sbase + (uint32_t)tid;
        const bool active = s < N;
        const uint64_t bits = bits_next;
        bits_next = s + BIG_T < N ? order[s + BIG_T] : 0ull;
        CumCounts cc = {0u, 0u, 0u};
        if (use_cnt) {
            /* the counts from the prefix sums: k + 1 differences per direction; a site assignment whose path crosses a
             * marked node is walked, looking its fragments at the marked nodes up itself */
            uint32_t marked = 0;
            if (active) cc = cnt_eval_sites(cntG, (uint32_t)bits, k, n_sites, &marked);
#ifdef PYA_STAMPS                                              /* ... site assignments / those with a marked node / wave rounds that walk (52-54) */
            if (b.stamps) {
                if (active) atomicAdd(&b.stamps[52], 1ull);
                if (marked != 0) atomicAdd(&b.stamps[53], 1ull);
                if (lane == 0 && __any(marked != 0)) atomicAdd(&b.stamps[54], 1ull);
                if (lane == 0) atomicAdd(&b.stamps[55], 1ull);
            }
#endif
            if (__any(marked != 0)) {
                /* (the residue mask only here: deposit_sites is a loop over the sites, 120 instructions a round that the
                 * table route needs once in a thousand rounds -- r06: it was computed ahead of the branch) */
                const uint64_t resmask = deposit_sites(bits, res.site_mask);
                float run0 = 0.f, run1 = 0.f;
                CumCounts cw = {0u, 0u, 0u};
                walk_cnt_both(env, tab, cum_lut, cnt_t, pos_cap, kc, resmask, 0, L - 1, run0, 0u, 0, L - 1, run1, 0u, cw);
                if (marked != 0) cc = cw;
            }
        } else {
            const uint64_t resmask = deposit_sites(bits, res.site_mask);
            const PrefixCompact p0 = l2[(uint32_t)(bits & 1023ull)];
            const PrefixCompact p1 = l2[1024u + (uint32_t)((__brevll(bits) >> (64 - n_sites)) & 1023ull)];
            float run0 = p0.running, run1 = p1.running;
            cc = entry_counts(p0);                               /* both directions' prefixes, then the rest of the walk */
            cc.add(make_uint4((uint32_t)p1.lo, (uint32_t)(p1.lo >> 32), p1.hi, 0u));
            walk_cum_both(env, tab, cum_lut, resmask, stop2[0], L - 1, run0, stop2[1], L - 1, run1, cc);
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
                    const float sc = lutl[cum[d] * PYA_NTOP + (uint32_t)d];
                    const float prod = cfg->weights[d] * sc;              /* float product ...   */
                    sum = sum + (double)prod;                             /* ... double sum      */
                }
                ws = (float)sum;
            } else {
                lut_fail = 1;
            }
            if (cand) wsl[s] = ws;
            else b.ws[s0 + s] = ws;
            const uint32_t u = __float_as_uint(ws);
            if (ws >= 0.f && (top_n == 0 || u > top_u)) {
                top_u = u;
                top_n = 1;
                top_i = s;
                top_b = (uint32_t)bits;
            } else if (ws >= 0.f && u == top_u) {
                top_n++;
            }
            if (b.rec && !inl) {
                uint32_t *rec = b.rec + (s0 + s) * PYA_REC_WORDS;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) rec[d >> 1] = cum[d] | (cum[d + 1] << 16);
                rec[5] = nfrag;
            }
        }
    }
    STAMP_T(b, 11, );
    /* (the PepScores are read back below by other wavefronts of this workgroup: the barriers' workgroup-scope
     * fences order that; an agent-scope fence here -- an L2 write-back per wavefront -- doubled the kernel's time) */
    /* ---- summary of the scores over the eight wavefronts ---- */
    {
        const uint32_t kmax = wave_max_u32(top_n ? top_u : 0u);
        const bool mine = top_n && top_u == kmax;
        const int n_max = wave_sum_i32(mine ? (int)top_n : 0);
        const uint32_t first = wave_min_u32(mine ? top_i : 0xffffffffu);
        const uint32_t first_b = wave_max_u32(mine && top_i == first ? top_b : 0u);
        const uint32_t failed_w = __any(lut_fail) ? 0x80000000u : 0u;
        if (lane == 0) {
            tops[wave * 4 + 0] = kmax;
            tops[wave * 4 + 1] = (uint32_t)n_max | failed_w;
            tops[wave * 4 + 2] = first;
            tops[wave * 4 + 3] = first_b;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t kmax = 0, n_max = 0, first = 0xffffffffu, first_b = 0, failed = 0;
        for (int wv = 0; wv < BIG_WAVES; wv++) {
            const uint32_t ku = tops[wv * 4], kn = tops[wv * 4 + 1] & 0x7fffffffu, kf = tops[wv * 4 + 2], kb = tops[wv * 4 + 3];
            failed |= tops[wv * 4 + 1] >> 31;
            if (kn == 0) continue;
            if (n_max == 0 || ku > kmax) {
                kmax = ku;
                n_max = kn;
                first = kf;
                first_b = kb;
            } else if (ku == kmax) {
                n_max += kn;
                first_b = kf < first ? kb : first_b;
                first = kf < first ? kf : first;
            }
        }
        uint32_t *t = b.ws_top + (size_t)psm * 4;
        t[0] = kmax;
        t[1] = n_max;
        t[2] = first;
        if (failed) b.status[psm] = PYA_ST_LUT_RANGE;
        tops[0] = kmax;                                      /* (every wavefront's entry has been read: the first one carries the summary on) */
        tops[1] = n_max;
        tops[2] = first;
        tops[3] = failed;
        tops[4] = first_b;
    }
    if (tid == 0) b.ws_top[(size_t)psm * 4 + 3] = 0u;            /* (candidate records: said below, once they exist) */
    if (!inl) return;
    /* ---------------- the winner, right here (see the note at the top) ---------------- */
    __syncthreads();
    const uint32_t kmax = tops[0], n_max = tops[1];
    if (tops[3] || n_max == 0) return;           /* (trial count outside the score table: localize writes "no result") */
    uint32_t best_i = tops[2];
    if (n_max != 1 || (b.debug & 1024u)) {
        /* a tie for the best PepScore: the front of std::sort decides (cpp/Ascore.cpp:141-146) */
        BigSortLds srt;
        unsigned char *aux;
        if (cand) {
            srt.key = wsl;                                       /* in place */
            srt.idx = (uint16_t *)((unsigned char *)l2 + g_bytes);
            srt.lq = srt.idx + N8;
            srt.rq = srt.lq + N8;
            aux = aux_in_tail ? (unsigned char *)lutl + row_bytes : (unsigned char *)(srt.rq + N8);
        } else {
            srt.key = (float *)l2;                               /* l2 | cum_lut: nothing reads them any more */
            srt.idx = (uint16_t *)(srt.key + N);
            srt.lq = srt.idx + N;
            srt.rq = srt.lq + N;
            aux = (unsigned char *)(((uintptr_t)(srt.rq + N) + 15) & ~(uintptr_t)15);
        }
        srt.ml = (unsigned long long *)aux;
        srt.mr = srt.ml + 64;
        srt.cl = (uint32_t *)(srt.mr + 64);
        srt.cr = srt.cl + 64;
        srt.misc = srt.cr + 64;
        for (uint32_t i = (uint32_t)tid; i < N; i += BIG_T) {
            if (!cand) srt.key[i] = b.ws[s0 + i];
            srt.idx[i] = (uint16_t)i;
        }
        __syncthreads();
        if (b.debug & 8u) return;
        bool ood;
        best_i = wg_spine_front(srt, (int)N, kmax, &ood);
        STAMP_T(b, 12, );
        if (ood) return;                                         /* (out of depth: the count stays, the localize kernel hands the PSM over) */
        if (tid == 0) {
            uint32_t *t = b.ws_top + (size_t)psm * 4;
            t[1] = 1u;
            t[2] = best_i;
        }
    }
    if (!cand || wave != 0) return;
    /* ---- the winner's single-move competitors (cpp/Ascore.cpp:212-254: item e = a * n_free + f moves the winner's a-th
     * modification to its f-th free site, the enumeration of localize_body) and, as item k (n - k), the winner itself:
     * counts from the per-site table, PepScore from the row, both left for the finishing kernel ---- */
    {
        /* (a unique best score: its thread kept the site assignment -- no dependent load from the order table) */
        const uint64_t best_bits = (n_max != 1 || (b.debug & 1024u)) ? order[best_i] : (uint64_t)tops[4];
        const uint64_t all_sites = (1ull << n_sites) - 1ull;     /* (n_sites <= 32 here) */
        const uint64_t free_bits = all_sites & ~best_bits;
        const FastDiv divF = fastdiv_make((uint32_t)n_free);
        float *out_ws = b.ws + s0;
        uint32_t *out_rec = (uint32_t *)out_ws + BIG_CAND_REC;
        for (int base = 0; base <= items; base += 64) {
            const int e = base + lane;
            const bool on = e <= items;
            uint64_t c = best_bits;
            if (on && e < items) {
                const int a = (int)fastdiv((uint32_t)e, divF), fb = e - a * n_free;
                c = (best_bits & ~(1ull << nth_set_bit(best_bits, a))) | (1ull << nth_set_bit(free_bits, fb));
            }
            uint32_t marked = 0;
            CumCounts cc = {0u, 0u, 0u};
            if (on) cc = cnt_eval_sites(cntG, (uint32_t)c, k, n_sites, &marked);
            if (__any(marked != 0)) {
                float run0 = 0.f, run1 = 0.f;
                CumCounts cw = {0u, 0u, 0u};
                walk_cnt_both(env, tab, cum_lut, cnt_t, pos_cap, kc, deposit_sites(c, res.site_mask), 0, L - 1, run0, 0u, 0, L - 1, run1, 0u, cw);
                if (marked != 0) cc = cw;
            }
            if (on) {
                uint32_t cum[PYA_NTOP];
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) cum[d] = cc.at(d);
                double sum = 0.;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) {
                    const float sc = lutl[cum[d] * PYA_NTOP + (uint32_t)d];
                    const float prod = cfg->weights[d] * sc;
                    sum = sum + (double)prod;
                }
                out_ws[e] = (float)sum;
                uint32_t *rec = out_rec + (size_t)e * PYA_REC_WORDS;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) rec[d >> 1] = cum[d] | (cum[d + 1] << 16);
                rec[5] = nfrag;
            }
        }
        if (lane == 0) b.ws_top[(size_t)psm * 4 + 3] = BIG_CAND_FLAG;
    }
}

__global__ __launch_bounds__(64 * BIG_WAVES, 6) void pya_score_big_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids,
                                                                       uint32_t cap, uint32_t pos_cap, uint32_t kc, BigLoc loc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    big_body(b, psm_ids[xcd_slot(blockIdx.x, n_ids)], lds_raw, cap, pos_cap, kc, loc);
}

/* the PSMs the in-kernel localisation declined, scored again with count records for the general localize body:
 * a small grid strides over the list */
__global__ __launch_bounds__(64 * BIG_WAVES, 6) void pya_score_big_list_kernel(BatchDev b, const uint32_t *count, const uint32_t *ids,
                                                                            uint32_t cap, uint32_t pos_cap, uint32_t kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t n = *count;
    BigLoc loc = {};
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        big_body(b, ids[k], lds_raw, cap, pos_cap, kc, loc);
        __syncthreads();
    }
}

extern "C" size_t pya_score_big_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc) { return score_big_lds_bytes(cap, pos_cap, kc); }

/* pos_cap: the largest L - 1 of the launch (sizes the score-table row kept in LDS).  inline_on: summary mode --
 * for PSMs of up to pya_big_inline_max() signatures no count records are written and a tie for the best score is
 * resolved in the kernel (pya_launch_localize_recount finishes them). */
/* kc: a power of two >= 8 and > the launch's largest number of modifications (the row length of the count-node table) */
extern "C" int pya_launch_score_big(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                                    uint32_t pos_cap, uint32_t kc, uint32_t inline_on, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = score_big_lds_bytes(cap, pos_cap, kc);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_score_big_kernel);
    if (e != hipSuccess) return (int)e;
    BigLoc loc = {inline_on};
    hipLaunchKernelGGL(pya_score_big_kernel, dim3(n_ids), dim3(64 * BIG_WAVES), lds, stream, *b, d_ids, n_ids, cap, pos_cap, kc, loc);
    return (int)hipGetLastError();
}

extern "C" int pya_launch_score_big_list(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max,
                                         uint32_t cap, uint32_t pos_cap, uint32_t kc, hipStream_t stream) {
    if (n_max == 0) return 0;
    const size_t lds = score_big_lds_bytes(cap, pos_cap, kc);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_score_big_list_kernel);
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = n_max < 2048u ? n_max : 2048u;
    hipLaunchKernelGGL(pya_score_big_list_kernel, dim3(grid), dim3(64 * BIG_WAVES), lds, stream, *b, d_count, d_ids, cap, pos_cap, kc);
    return (int)hipGetLastError();
}
