/* localize_body.hip.h -- localisation of one PSM by one wavefront (both instantiations); see
 * rank_and_localize.hip for the notes.  Shared with tiny_batch.hip. */
#ifndef PYA_LOCALIZE_BODY_H
#define PYA_LOCALIZE_BODY_H
#include "device_common.hip.h"
#include "localize_core.hip.h"

/* the hash route: same carve-up with the fragment pool and its keep flags replaced by loc_hash_words() words */
static inline size_t localize_hash_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc,
                                             uint32_t hs, uint32_t pp, uint32_t max_k, uint32_t n_nl);
static inline size_t localize_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap,
                                        uint32_t sb) {
    /* (the localize kernel looks peaks up in global memory: no peak table here) */
    size_t fixed = 512 + PYA_MAX_UNIQ * 4 + (size_t)push_cap * 16 + 64 * 16 + 16;
    size_t srt = n_cap ? sort_lds_bytes(n_cap) + 64 : 64;
    size_t lst = pya_loc_lds_bytes(pos_cap, pool_cap, sb);
    return fixed + (srt > lst ? srt : lst) + 64;
}

/* the hash route sizes the per-site and per-residue arrays by the launch (its LDS decides its occupancy) */
static inline __host__ __device__ uint32_t hash_site_cap(uint32_t max_k) {
    const uint32_t v = (max_k + 3u) & ~3u;
    return v > 64u ? 64u : (v < 4u ? 4u : v);
}
/* entries of the staged neutral-loss state table: the state packs two bits per distinct loss mass */
static inline __host__ __device__ uint32_t hash_nl_cap(uint32_t n_nl) { return n_nl >= 4u ? 256u : (n_nl == 0u ? 4u : 1u << (2u * n_nl)); }
static inline __host__ __device__ uint32_t hash_res_cap(uint32_t pos_cap) {
    const uint32_t v = (pos_cap + 1u + 3u) & ~3u;
    return v > 64u ? 64u : v;
}
static inline size_t localize_hash_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc,
                                             uint32_t hs, uint32_t pp, uint32_t max_k, uint32_t n_nl) {
    const uint32_t site_cap = hash_site_cap(max_k), res_cap = hash_res_cap(pos_cap);
    size_t fixed = 2 * (size_t)hash_nl_cap(n_nl) + PYA_MAX_UNIQ * 4 + (size_t)push_cap * 16 + (size_t)site_cap * 16 + 16;
    size_t srt = n_cap ? sort_lds_bytes(n_cap) + 64 : 64;
    /* (residue arrays by the launch, no span tables, 128 one-byte staging tags instead of 128 words) */
    size_t lst = pya_loc_lds_bytes(pos_cap, 0, sb) - (64 - res_cap) * 9 - (32 + 33) * 4 - (128 - 32) * 4 + 4 * loc_hash_words(vc, hs, pp, sb);
    return fixed + (srt > lst ? srt : lst) + 64;
}

/* r06: the lean instantiation's own size -- what it never touches is left out: the loss-variant tables of the run tables
 * (a quarter of their bytes), the staged loss-state table (512 B), per-site and per-residue arrays beyond the launch's
 * largest k and longest peptide.  10.6 -> 7.6 KB on cfg5's finishing launch, 15 -> 21 wavefronts per CU: these kernels wait
 * for memory (valu_busy 0.36), residency is what hides it. */
static inline size_t localize_lean_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                                             uint32_t max_k) {
    const uint32_t site_cap = hash_site_cap(max_k), res_cap = hash_res_cap(pos_cap);
    size_t fixed = 2 * (size_t)hash_nl_cap(0) + PYA_MAX_UNIQ * 4 + (size_t)push_cap * 16 + (size_t)site_cap * 16 + 16;
    size_t srt = n_cap ? sort_lds_bytes(n_cap) + 64 : 64;
    size_t lst = pya_loc_lds_bytes(pos_cap, pool_cap, sb) - (64 - res_cap) * 8 - (size_t)sb * 2 * pos_cap * 4;
    return fixed + (srt > lst ? srt : lst) + 64;
}

/* (five wavefronts per SIMD -- 96 registers, ~60 spilled -- and the LDS trimmed to 8 KB to match were measured:
 * 10.8-11.7 ms on cfg4 against 10.4 with four and no spills) */
#ifndef LOC_WAVES_HASH
#define LOC_WAVES_HASH 4
#endif
#ifndef LOC_WAVES
#define LOC_WAVES 4            /* general instantiation: 128 VGPRs */
#endif
#ifndef LOC_WAVES_PLAIN
#define LOC_WAVES_PLAIN 5      /* lean instantiation: fits 102 VGPRs without scratch */
#endif

/* One PSM, one wavefront.  Two instantiations:
 *   PLAIN = true  -- no neutral losses, fragment charge 1, every residue mass positive (so every
 *                    fragment list is ascending and position-indexed), summary mode.  Everything
 *                    only other PSMs need is compiled out, which takes the kernel from 128 to
 *                    under 102 VGPRs = 5 instead of 4 waves per SIMD (this kernel's speed follows
 *                    its occupancy).  Returns true -- nothing written -- when the PSM turns out to
 *                    need a route it does not have: an ion with two partners within mz_error,
 *                    introsort running out of depth, a non-positive residue mass.  Such PSMs are appended to a list and redone by
 *                    the general instantiation.
 *   PLAIN = false -- everything. */
/* What a caller that has scored the PSM itself hands over instead of the workspace arrays (score_big's own
 * localisation): the PepScores in LDS, the winner (the std::sort front it has already determined), the peak
 * table and grid it has staged in LDS, and room for the recounted records of a batch of signatures. */
struct InlineSrc {
    const float *ws;          /* [N] PepScores, pre-sort order (LDS) */
    uint32_t best_i, kmax;    /* winner's pre-sort index, its PepScore (bits) */
    PeakTable tab;            /* staged table (tab.e, tab.cell set) */
    uint32_t *rec_batch;      /* [sb][PYA_REC_WORDS] */
    uint32_t *hist;           /* [sb][PYA_NTOP] */
    bool valid;               /* false (the default): nothing handed in */
    bool cand;                /* r06: `ws` holds the PepScores of the winner's k (n - k) single-move competitors in the body's own
                               * item order and the winner's behind them, count records from word PYA_CAND_REC on (score_big's
                               * candidate mode): no combination ranks, no gather, no recount */
};
#define PYA_CAND_REC 128

/* HASH (general instantiation only): the site-determining ions come from loc_site_ions_hash; pool_cap is then the
 * number of 4-byte words of its work area (loc_hash_words(vc, hs, pp)), and a PSM it declines returns true. */
template <bool PLAIN, bool HASH = false>
DEV bool localize_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t push_cap, uint32_t pos_cap,
                       uint32_t pool_cap, uint32_t sb, uint32_t gtp, bool sort_room = true, const InlineSrc in = InlineSrc(),
                       uint32_t vc = 0, uint32_t hs = 0, uint32_t pp = 0) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    /* (r06: the prologue's loads in two rounds -- device_common.hip.h: load_desc) */
    LetterRegs letters = {0.f, 0u};
    if (!HASH) letters = load_letter_regs(cfg);             /* (the hash route is short of registers: two more live ones cost it ten spilled -- it keeps load_residues) */
    const PsmDesc dsc = load_desc(b, psm);
    const int status0 = b.status[psm];
    const int R0 = (int)b.ret_n[psm];
    /* (the score summary beside them: read further down, it was a round trip of its own behind the status test) */
    uint4 top4 = make_uint4(0u, 0u, 0u, 0u);
    if (!in.valid) top4 = *(const uint4 *)(b.ws_top + (size_t)psm * 4);
    const int k = dsc.k;
    const uint32_t max_k = b.max_k;
    float *out_asc = b.ascores + (size_t)psm * max_k;
    uint64_t *out_alt = b.alt_mask + (size_t)psm * max_k;

    for (uint32_t a = lane; a < max_k; a += 64) {
        out_asc[a] = 0.f;
        out_alt[a] = 0ull;
    }
    if (status0 != PYA_ST_OK) {
        if (lane == 0) {
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return false;
    }

    const int N = (int)dsc.N;
    const int n_sites = dsc.n_sites;
    const uint64_t *order = b.order_tab + dsc.order_off;
    const int64_t s0 = dsc.sig0;
    const bool use_in = in.valid;
    const float *ws = use_in ? in.ws : b.ws + s0;

    /* Ascore::isUnambiguous, cpp/Ascore.cpp:38-51 */
    if (k >= n_sites) {
        for (int a = lane; a < k && a < (int)max_k; a += 64) out_asc[a] = __builtin_huge_valf();
        if (lane == 0) {
            b.best_score[psm] = N > 0 ? ws[0] : -1.f;
            b.best_sig[psm] = N > 0 ? order[0] : 0ull;
            b.n_sig_out[psm] = N;
            if (b.keep && N > 0) b.sorted_idx[s0] = 0;
        }
        return false;
    }

    STAMP_BEGIN();
    /* r06: the binomials the competitor enumeration ranks combinations with -- C(site, t), t = 1 .. k -- requested here, while
     * the tables below are staged, and parked in LDS before the enumeration: its rank loop made k global loads one after the
     * other (a per-lane trip count: each load waited for), then the dependent loads of the order table and the PepScore --
     * six round trips where three remain (r06 stamps: the enumeration was a fifth of the lean kernel's wave time on cfg3) */
    const int nk = n_sites * k;
    const bool binom_lds = !(use_in && in.cand) && nk <= 128;
    uint32_t bin_pre[2] = {0u, 0u};
    if (binom_lds) {
        const FastDiv divS = fastdiv_make((uint32_t)n_sites);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int i = r * 64 + lane;
            if (i < nk) {
                const int t1 = (int)fastdiv((uint32_t)i, divS);       /* t - 1 */
                bin_pre[r] = b.binom[(i - t1 * n_sites) * 64 + t1 + 1];
            }
        }
    }
    /* only a handful of ions are matched here, so the retained-peak table is not staged in LDS:
     * that keeps this kernel's LDS small (occupancy) and saves the staging + grid build */
    /* (the hash route looks up ~1 300 ions per PSM on cfg4-like settings; staging the table for it was measured
     * slower -- occupancy -- and is gone: DESIGN.md section 10) */
    const uint32_t site_cap = (HASH || PLAIN) ? hash_site_cap(max_k) : 64u, res_cap = (HASH || PLAIN) ? hash_res_cap(pos_cap) : 64u;
    K3Lds lds = carve(lds_raw, 0, false, push_cap, site_cap, HASH ? hash_nl_cap((uint32_t)cfg->n_nl) : (PLAIN ? hash_nl_cap(0u) : 256u));
    LocCtx ctx;
    ctx.b = &b;
    ctx.cfg = cfg;
    stage_tables(b, cfg, lds, psm, &ctx.tab, &ctx.nl, false, dsc.ret0, R0);
    if (use_in) ctx.tab = in.tab;                               /* (the caller's table in LDS: lookups stay on chip) */
    const Residues res = HASH ? load_residues(b, cfg, psm) : load_residues_desc(b, cfg, dsc, letters);
    const uint64_t site_mask_u = res.site_mask;
    /* positive residue masses make the float32 running sum, hence every m/z list, ascending */
    const int zmax = dsc.zmax;
    const bool presorted = zmax == 1 && cfg->n_nl == 0 &&
                           !__any(lane < res.L && !(res.m0 > 0.f && res.m1 > 0.f));
    const float wide_min = 2.f * cfg->mz_error + 0.02f;
    const bool wide = presorted && !__any(lane < res.L && !(res.m0 > wide_min && res.m1 > wide_min)) && !(b.debug & 2048);
    if (PLAIN && (!presorted || b.keep || (b.debug & 512))) return true;   /* not this kernel's PSM */
    STAMP_T(b, 20, false);

    /* ---- sort (cpp/Ascore.cpp:141-146) ---- */
    if (lane == 0) *lds.n_pushed = 0;
    if ((uint32_t)lane < site_cap) {
        lds.site_max[lane] = 0;
        lds.site_tie[lane] = 0;
        lds.site_alt[lane] = 0ull;
    }
    /* The winner is the front of the sorted list: the largest PepScore, and among equal ones
     * whichever std::sort leaves first.  When the maximum is unique (4 PSMs in 5) no emulation is
     * needed to name it. */
    const float ws_lane = lane < N ? ws[lane] : 0.f;       /* the first 64 scores stay in a register */
    uint32_t kmax = 0, first_max = 0xffffffffu;
    int n_max = 0;
    if (use_in) {                                           /* the caller knows the winner already */
        kmax = in.kmax;
        n_max = 1;
        first_max = in.best_i;
    } else {
        kmax = top4.x;                                      /* score_signatures' summary (0 signatures = none) */
        n_max = (int)top4.y;
        first_max = top4.z;
    }
    if (n_max == 0) {
        kmax = 0;
        first_max = 0xffffffffu;
        for (int i = lane; i < N; i += 64) {
            const uint32_t u = __float_as_uint(i < 64 ? ws_lane : ws[i]);   /* scores are >= 0: bit order = value order */
            kmax = u > kmax ? u : kmax;
        }
        kmax = wave_max_u32(kmax);
        for (int i = lane; i < N; i += 64) {
            if (__float_as_uint(i < 64 ? ws_lane : ws[i]) == kmax) {
                n_max++;
                first_max = first_max < (uint32_t)i ? first_max : (uint32_t)i;
            }
        }
        n_max = wave_sum_i32(n_max);
        first_max = wave_min_u32(first_max);
    }
    uint32_t best_i = first_max;
    STAMP_T(b, 21, false);
    /* The sort emulation needs 10 bytes of LDS per signature, which for thousands of signatures is
     * what decides this kernel's occupancy -- and most PSMs have a unique best PepScore and never
     * sort.  Big-C(n,k) launches of the lean instantiation therefore run without that room and hand the
     * PSMs with a tie at the top to the general instantiation. */
    if (PLAIN && !sort_room && !use_in && (n_max != 1 || (b.debug & 1024))) return true;
    if (!use_in && (n_max != 1 || b.keep || (b.debug & 1024))) {
        const SortLds srt = sort_carve(lds.scratch, N);
        /* (eight loads on their way before the first is stored: thousands of scores, and a wavefront
         * that makes one memory round trip per 64 of them spends its time waiting) */
        for (int base = 0; base < N; base += 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = base + u * 64 + lane;
                v[u] = ws[i < N ? i : N - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int i = base + u * 64 + lane;
                if (i < N) {
                    srt.key[i] = v[u];
                    srt.idx[i] = (uint16_t)i;
                }
            }
        }
        wave_lds_sync();
        /* only the left spine of the partition tree decides the front element; the full sort is
         * needed when the caller wants the whole ordering */
        int front_len = N;
        if (!(b.debug & 8)) {
            if (sort_introsort_loop<PLAIN>(srt, N, b.keep == 0, &front_len)) return true;
        }
        /* front of the sorted list = left-most maximum of the partitioned array (which the spine's last,
         * left-most run holds: every element of it is at least as large as anything to its right) */
        uint32_t first_pos = 0xffffffffu;
        for (int i = lane; i < front_len; i += 64)
            if (__float_as_uint(srt.key[i]) == kmax) first_pos = first_pos < (uint32_t)i ? first_pos : (uint32_t)i;
        first_pos = wave_min_u32(first_pos);
        best_i = srt.idx[first_pos];
        if (b.keep) {
            for (int i = lane; i < N; i += 64) b.sorted_idx[s0 + sort_final_pos(srt, i, N)] = srt.idx[i];
        }
    }
    STAMP_T(b, 22, false);
    /* (wave-uniform values as scalars: they stay live across everything below, and a vector register holds 64 copies) */
    best_i = (uint32_t)__builtin_amdgcn_readfirstlane((int)best_i);
    const float best_ws = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)kmax));
    const uint64_t best_bits_v = order[best_i];
    const uint64_t best_bits = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)best_bits_v) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(best_bits_v >> 32)) << 32);
    wave_lds_sync();

    STAMP_T(b, 23, false);
    /* ---- single-move competitors (cpp/Ascore.cpp:212-254).  The reference finds them by scanning the
     * sorted list; they are exactly the k * (n - k) signatures that differ from the winner by one moved
     * modification, so they are enumerated directly: signature -> combination rank (sum of C(p_t, t)
     * over its set bits) -> its index in the pre-sort order (inv_tab) -> its PepScore.  One item per
     * (modified site, free site) pair instead of two passes over all C(n,k) signatures. ---- */
    {
        const uint64_t all_sites = n_sites >= 64 ? ~0ull : ((1ull << n_sites) - 1ull);
        const uint64_t free_bits = all_sites & ~best_bits;
        const int n_free = n_sites - k, items = k * n_free;
        const uint32_t *inv = b.inv_tab + dsc.order_off;
        const FastDiv divF = fastdiv_make((uint32_t)(n_free > 0 ? n_free : 1));
        uint32_t *bl = (uint32_t *)lds.scratch;              /* [k][n_sites] (the sort is over, the batch tables not yet carved) */
        if (binom_lds) {
#pragma unroll
            for (int r = 0; r < 2; r++)
                if (r * 64 + lane < nk) bl[r * 64 + lane] = bin_pre[r];
            wave_lds_sync();
        }
        /* items <= 126 (PYA_MAX_PUSHED): two per lane at most, both kept in registers between the passes */
        uint64_t c_r[2] = {0ull, 0ull};
        int a_r[2] = {0, 0};
        uint32_t u_r[2] = {0u, 0u}, idx_r[2] = {0u, 0u};
        bool on_r[2] = {false, false};
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int e = r * 64 + lane;
            on_r[r] = e < items;
            if (on_r[r]) {
                const int a = (int)fastdiv((uint32_t)e, divF);
                const int fb = e - a * n_free;
                const int pos_a = nth_set_bit(best_bits, a), pos_b = nth_set_bit(free_bits, fb);
                const uint64_t c = (best_bits & ~(1ull << pos_a)) | (1ull << pos_b);
                uint32_t idx = (uint32_t)e;                    /* (candidate records: the item number is the index) */
                if (!in.cand) {
                    uint32_t rank = 0;
                    uint64_t m = c;
                    for (int t = 1; m; t++) {                  /* colexicographic rank of the combination */
                        const int pos = __builtin_ctzll(m);
                        m &= m - 1;
                        rank += binom_lds ? bl[(t - 1) * n_sites + pos] : b.binom[pos * 64 + t];
                    }
                    idx = inv[rank];
                }
                const uint32_t u = __float_as_uint(ws[idx]);
                atomicMax(&lds.site_max[a], u);
                c_r[r] = c;
                a_r[r] = a;
                u_r[r] = u;
                idx_r[r] = idx;
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (on_r[r] && u_r[r] == lds.site_max[a_r[r]]) {
                if ((double)__builtin_fabsf(best_ws - __uint_as_float(u_r[r])) < 1e-6) {
                    /* ties the winner: Ascore 0 (Ascore.cpp:159-161), no ion work needed */
                    lds.site_tie[a_r[r]] = 1u;
                    atomicOr(&lds.site_alt[a_r[r]], 1ull << nth_set_bit(site_mask_u, __builtin_ctzll(c_r[r] & ~best_bits)));
                } else {
                    const uint32_t slot = atomicAdd(lds.n_pushed, 1u);
                    if (slot < push_cap) {
                        PushedEntry pe;
                        pe.bits = c_r[r];
                        pe.ws = __uint_as_float(u_r[r]);
                        pe.idx = idx_r[r];
                        lds.pushed[slot] = pe;
                    }
                }
            }
        }
        wave_lds_sync();
    }
    const uint32_t n_pushed = (uint32_t)__builtin_amdgcn_readfirstlane((int)*lds.n_pushed);
    int fail = 0;
    if (n_pushed > push_cap) fail = 2;                    /* cannot happen: push_cap >= k * (n_sites - k) */
    uint32_t np = n_pushed < push_cap ? n_pushed : push_cap;
    if (b.debug & 16) np = 0;

    STAMP_T(b, 24, false);
    /* ---- Ascores, sb-1 competitors at a time ---- */
    ctx.w = loc_carve(lds.scratch, pos_cap, pool_cap, sb, res_cap, !HASH, !PLAIN);
    ctx.sb = (int)sb;
    ctx.gtp = (int)gtp;
    ctx.L = res.L;
    ctx.zmax = zmax;
    ctx.presorted = presorted;
    ctx.wide = wide;
    ctx.pos_cap = pos_cap;
    ctx.pool_cap = pool_cap;
    const LocLds &w = ctx.w;
    if ((uint32_t)lane < res_cap) {
        w.m0[lane] = res.m0;
        w.m1[lane] = res.m1;
        w.nlp[lane] = (uint8_t)res.nl;
    }
    if (lane == 0) w.sig_mask[0] = deposit_sites(best_bits, res.site_mask);
    wave_lds_sync();

    STAMP_T(b, 25, false);
    float my_asc = __builtin_huge_valf();     /* lane a keeps site a */
    uint64_t my_alt = 0ull;
    HashLds hl;
    if (HASH) hl = hash_carve(w.pool, vc, hs, pp, sb);
    const bool cand = use_in && in.cand;
    const bool declined = loc_ascore_all<PLAIN, HASH>(ctx, lds.pushed, np, lds.site_alt,
                   cand ? (const uint32_t *)ws + PYA_CAND_REC : b.rec + s0 * PYA_REC_WORDS,
                   best_bits, best_ws, cand ? (uint32_t)(k * (n_sites - k)) : best_i, res.site_mask,
                   &my_asc, &my_alt, &fail, (use_in && !cand) ? in.rec_batch : nullptr, (use_in && !cand) ? in.hist : nullptr, HASH ? &hl : nullptr);
    if ((PLAIN || HASH) && declined) return true;
    STAMP_T(b, 36, false);
    if (lane < k && lds.site_tie[lane]) my_asc = 0.f < my_asc ? 0.f : my_asc;
    if (lane < k) my_alt |= lds.site_alt[lane];
    if (lane < k && lane < (int)max_k) {
        out_asc[lane] = my_asc;
        out_alt[lane] = my_alt;
    }
    const bool any_fail = __any(fail != 0), overflow = __any(fail == 2);
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (any_fail) b.status[psm] = overflow ? PYA_ST_PUSHED_OVERFLOW : PYA_ST_LUT_RANGE;
    }
    return false;
}

#endif
