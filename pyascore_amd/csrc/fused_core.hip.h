/* fused_core.hip.h -- scoring AND localisation of one PSM by one wavefront, in one pass, for the
 * PSMs that make up most real batches: few site assignments (C(n,k) <= 32, or <= 64 with ion types
 * of one direction only), no neutral losses, fragment charge 1, one ion type per direction.
 *
 * Replaces, for those PSMs, score_signatures + rank_and_localize (cpp/Ascore.cpp:53-254,
 * cpp/ModifiedPeptide.cpp:126-150, :259-320, :326-609).  The two-kernel route makes every walker
 * throw away what localisation needs again -- each fragment's m/z and the rank of the peak it
 * matched -- and then re-derives it: per-signature prefix tables walked by a handful of lanes,
 * fragment lists rebuilt from them, every surviving ion looked up in the peak table in global
 * memory, the per-signature scores, counts and the m/z grid written to HBM by one kernel and read
 * back by the next.  Here every (signature, direction) walker records its L-1 fragment m/z and
 * matched ranks in LDS while it scores (two stores per step), so that once the winner and its
 * single-move competitors are known the site-determining ions are a comparison of recorded lists
 * and a count over recorded ranks: no second walk, no lookup, nothing through HBM but the inputs
 * and the 64-byte result.
 *
 * Exactness is that of the two kernels (same walker, same window test, same std::sort emulation,
 * same neighbour-probe pairing): a PSM that needs a route this body does not have -- a residue mass
 * that is not positive (lists not ascending), an ion with two partners within mz_error, introsort
 * running out of depth -- is handed over: its scores, count records and grid are written where
 * score_signatures would have left them and the general localize instantiation redoes it.
 */
#ifndef PYA_FUSED_CORE_H
#define PYA_FUSED_CORE_H
#include "score_core.hip.h"
#include "localize_core.hip.h"

/* LDS of one wavefront:
 *   walk region  grid u16[256] | cnt u32[5][64] | resd float2[64] | peaks PeakEntry[cap + 4]
 *   lists        mzl f32[pos_cap][stride] | rkl u8[pos_cap][stride]      stride = walkers, rounded up to 4
 *   records      rec6 u32[n_cap][6] | wsl f32[n_cap]
 *   post region  (aliases cnt | resd | peaks after the walk) sort arrays, pushed competitors, per-site
 *                maxima / ties / alternative sites, per-competitor depth scores and counters            */
struct FusedLds {
    uint16_t *grid;
    uint32_t *cnt;
    float2 *resd;
    PeakEntry *peaks;
    float *mzl;
    uint8_t *rkl;
    uint32_t *rec6;
    float *wsl;
    /* post region */
    float *sort_key;
    uint16_t *sort_idx, *sort_l, *sort_r;
    PushedEntry *pushed;
    unsigned long long *site_alt;
    uint32_t *site_max, *site_tie, *n_pushed;
    float *sc;               /* [(1 + push_cap)][10] depth scores: winner, then competitors */
    uint32_t *c_tr, *c_cnt;  /* [push_cap][2] */
    int32_t *c_depth;        /* [push_cap] */
    uint32_t *c_site;        /* [push_cap] */
};

__host__ __device__ static inline size_t fused_post_bytes(uint32_t n_cap, uint32_t push_cap) {
    return (size_t)n_cap * 10 + 16 + (size_t)push_cap * 16 + 64 * 8 + 64 * 4 * 2 + 16 +
           (size_t)(1 + push_cap) * 40 + (size_t)push_cap * (8 + 8 + 4 + 4) + 64;
}
__host__ __device__ static inline size_t fused_walk_bytes(uint32_t cap) {
    return PYA_GRID_CELLS * 2 + PYA_NTOP / 2 * 64 * 4 + 64 * 8 + ((size_t)cap + PYA_TABLE_PAD) * 8;
}
static inline size_t fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t push_cap) {
    const size_t walk = fused_walk_bytes(cap), post = PYA_GRID_CELLS * 2 + fused_post_bytes(n_cap, push_cap);
    return (walk > post ? walk : post) + (size_t)pos_cap * stride * 5 + 16 + (size_t)n_cap * 28 + 64;
}

DEV FusedLds fused_carve(unsigned char *raw, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
                         uint32_t push_cap) {
    FusedLds f;
    f.grid = (uint16_t *)raw;
    f.cnt = (uint32_t *)(f.grid + PYA_GRID_CELLS);
    f.resd = (float2 *)(f.cnt + PYA_NTOP / 2 * 64);
    f.peaks = (PeakEntry *)(f.resd + 64);
    /* post region: after the grid (which a handed-over PSM still has to write out) */
    unsigned char *p = (unsigned char *)f.cnt;
    f.sort_key = (float *)p;
    f.sort_idx = (uint16_t *)(f.sort_key + n_cap);
    f.sort_l = f.sort_idx + n_cap;
    f.sort_r = f.sort_l + n_cap;
    p = (unsigned char *)(((uintptr_t)(f.sort_r + n_cap) + 15) & ~(uintptr_t)15);
    f.pushed = (PushedEntry *)p;
    f.site_alt = (unsigned long long *)(f.pushed + push_cap);
    f.site_max = (uint32_t *)(f.site_alt + 64);
    f.site_tie = f.site_max + 64;
    f.n_pushed = f.site_tie + 64;
    f.sc = (float *)(f.n_pushed + 4);
    f.c_tr = (uint32_t *)(f.sc + (size_t)(1 + push_cap) * 10);
    f.c_cnt = f.c_tr + 2 * push_cap;
    f.c_depth = (int32_t *)(f.c_cnt + 2 * push_cap);
    f.c_site = (uint32_t *)(f.c_depth + push_cap);
    const size_t walk = fused_walk_bytes(cap), post = PYA_GRID_CELLS * 2 + fused_post_bytes(n_cap, push_cap);
    unsigned char *tail = raw + (((walk > post ? walk : post) + 15) & ~(size_t)15);
    f.mzl = (float *)tail;
    f.rkl = (uint8_t *)(f.mzl + (size_t)pos_cap * stride);
    f.rec6 = (uint32_t *)(((uintptr_t)(f.rkl + (size_t)pos_cap * stride) + 15) & ~(uintptr_t)15);
    f.wsl = (float *)(f.rec6 + (size_t)n_cap * PYA_REC_WORDS);
    return f;
}

/* the straight-line walker of walk_core.hip.h that also records every fragment's m/z and matched
 * rank in column `w` of the lists */
DEV void walk_record(const WalkEnv &e, const PeakTable &tab, uint64_t resmask, int dir, bool active, float *mzl,
                     uint8_t *rkl, int stride, int w) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
    const float2 *rp = e.resd + (dir ? L - 1 : 0);
    const int rstride = dir ? -1 : 1;
    uint32_t *col = e.cnt + lane_id();
    float *mo = mzl + w;
    uint8_t *ro = rkl + w;
    float running = 0.f;
    for (int step = 0; step + 1 < L; step++, rp += rstride, mo += stride, ro += stride) {
        const float2 mm = *rp;
        const uint32_t word = step < 32 ? tlo : thi;
        const bool mod = (word >> (step & 31)) & 1u;
        const float r = mod ? mm.y : mm.x;
        running = r + running;                             /* ModifiedPeptide.cpp:385-389 */
        const double m = ((double)running + A) - B;
        const float f = (float)(m + 1.007825);
        const int rk = match_rank_lds(tab, f);
        hist_bump(col, active, rk);
        if (active) {
            *mo = f;
            *ro = (uint8_t)rk;
        }
    }
}

/* BOTH: ion types of both directions -- lanes 0..31 walk from the N-terminus, lanes 32..63 from the
 * C-terminus (C(n,k) <= 32); otherwise one direction, one signature per lane (C(n,k) <= 64).
 * Returns true when the PSM was handed over to the general localize instantiation. */
template <bool BOTH>
DEV bool fused_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t n_cap, uint32_t stride,
                    uint32_t pos_cap, uint32_t push_cap) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const uint32_t max_k = b.max_k;
    float *out_asc = b.ascores + (size_t)psm * max_k;
    uint64_t *out_alt = b.alt_mask + (size_t)psm * max_k;
    for (uint32_t a = lane; a < max_k; a += 64) {
        out_asc[a] = 0.f;
        out_alt[a] = 0ull;
    }
    if (b.status[psm] != PYA_ST_OK) {
        if (lane == 0) {
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return false;
    }
    const FusedLds f = fused_carve(lds_raw, cap, n_cap, stride, pos_cap, push_cap);
    const int N = (int)b.n_sig[psm];
    const int k = b.n_of_mod[psm];
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];

    /* ---- score: score_core.hip.h's small-C(n,k) route with recording walkers ---- */
    const Residues res = load_residues(b, cfg, psm);
    PeakTable tab;
    stage_peak_table(b, psm, f.peaks, &tab);
    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = 0;
    env.nl_present = nullptr;
    env.nl_uniq = nullptr;
    env.resd = f.resd;
    env.resn = nullptr;
    env.cnt = f.cnt;
    env.L = res.L;
    env.zmax = 1;
    stage_residues(res, f.resd, nullptr);
    wave_lds_sync();
    grid_build(&tab, f.grid);
    hist_clear(env);
    const int Lm1 = res.L - 1;
    const int s = BOTH ? (lane & 31) : lane;
    const int dir = BOTH ? (lane >> 5) : (cfg->n_fwd > 0 ? 0 : 1);
    const bool active = s < N;
    const uint64_t bits = active ? order[s] : 0ull;
    const uint64_t resmask = deposit_sites(bits, res.site_mask);
    const int w = BOTH ? (lane >> 5) * N + s : s;          /* this lane's column of the lists */
    /* positive residue masses make every list ascending (what the neighbour-probe pairing needs) */
    const bool presorted = !__any(lane < res.L && !(res.m0 > 0.f && res.m1 > 0.f));
    wave_lds_sync();
    walk_record(env, tab, resmask, dir, active, f.mzl, f.rkl, (int)stride, w);
    wave_lds_sync();

    const uint32_t nfrag = (BOTH ? 2u : 1u) * (uint32_t)Lm1;
    int fail = 0;
    float ws = 0.f;
    if (active && (!BOTH || lane < 32)) {
        /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
        uint32_t cum[PYA_NTOP];
        uint32_t acc = 0;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d++) {
            acc += hist_count(f.cnt, lane, d) + (BOTH ? hist_count(f.cnt, lane + 32, d) : 0u);
            cum[d] = acc;
        }
        ws = -1.f;
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
            fail = 1;
        }
        uint32_t *r6 = f.rec6 + (size_t)s * PYA_REC_WORDS;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d += 2) r6[d >> 1] = cum[d] | (cum[d + 1] << 16);
        r6[5] = nfrag;
        f.wsl[s] = ws;
    }
    if (__any(fail)) {                                      /* trial count outside the score table */
        if (lane == 0) {
            b.status[psm] = PYA_ST_LUT_RANGE;
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return false;
    }
    wave_lds_sync();                                        /* cnt / resd / peaks are free from here on */

    bool declined = !presorted || (b.debug & 512);
    const bool sig_lane = lane < N;                         /* lane i <-> signature i from here on */
    const uint64_t my_bits = bits;                          /* (lanes 32.. hold copies of 0..31's) */
    const float my_ws = sig_lane ? f.wsl[lane] : 0.f;
    float best_ws = 0.f;
    uint64_t best_bits = 0ull;
    uint32_t best_i = 0, np = 0;
    if (!declined) {
        /* ---- winner: the front of std::sort (cpp/Ascore.cpp:141-146) ---- */
        const uint32_t u = sig_lane ? __float_as_uint(my_ws) : 0u;        /* scores are >= 0: bit order = value order */
        const uint32_t kmax = wave_max_u32(u);
        const uint64_t at_max = __ballot(sig_lane && u == kmax);
        best_i = (uint32_t)__builtin_ctzll(at_max);
        if (lane == 0) *f.n_pushed = 0;
        f.site_max[lane] = 0;
        f.site_tie[lane] = 0;
        f.site_alt[lane] = 0ull;
        if (__popcll(at_max) != 1 || (b.debug & 1024)) {
            SortLds srt;
            srt.key = f.sort_key;
            srt.idx = f.sort_idx;
            srt.lpos = f.sort_l;
            srt.rpos = f.sort_r;
            if (sig_lane) {
                srt.key[lane] = my_ws;
                srt.idx[lane] = (uint16_t)lane;
            }
            wave_lds_sync();
            if (!(b.debug & 8) && sort_introsort_loop<true>(srt, N, true)) declined = true;
            const uint64_t m2 = __ballot(sig_lane && __float_as_uint(srt.key[lane]) == kmax);
            best_i = srt.idx[__builtin_ctzll(m2)];
        }
        best_ws = __uint_as_float(kmax);
        best_bits = __shfl(my_bits, (int)best_i, 64);
        wave_lds_sync();
    }
    if (!declined) {
        /* ---- single-move competitors (cpp/Ascore.cpp:212-254) ---- */
        const uint64_t gone = best_bits & ~my_bits, came = my_bits & ~best_bits;
        const bool single = sig_lane && __popcll(gone) == 1 && __popcll(came) == 1;
        const int a = single ? __popcll(best_bits & (gone - 1)) : 0;
        const uint32_t u = __float_as_uint(my_ws);
        if (single) atomicMax(&f.site_max[a], u);
        wave_lds_sync();
        if (single && u == f.site_max[a]) {
            if ((double)__builtin_fabsf(best_ws - my_ws) < 1e-6) {
                /* ties the winner: Ascore 0 (Ascore.cpp:159-161), no ion work needed */
                f.site_tie[a] = 1u;
                atomicOr(&f.site_alt[a], 1ull << nth_set_bit(res.site_mask, __builtin_ctzll(came)));
            } else {
                const uint32_t slot = atomicAdd(f.n_pushed, 1u);
                if (slot < push_cap) {
                    PushedEntry pe;
                    pe.bits = my_bits;
                    pe.ws = my_ws;
                    pe.idx = (uint32_t)lane;
                    f.pushed[slot] = pe;
                }
            }
        }
        wave_lds_sync();
        np = *f.n_pushed;
        if (np > push_cap) np = push_cap;                   /* cannot happen: push_cap >= k * (n_sites - k) */
        if (b.debug & 16) np = 0;
    }
    float my_asc = __builtin_huge_valf();                   /* lane a keeps site a */
    if (!declined && np > 0) {
        /* ---- depth scores of the winner and the competitors, read off the score table ---- */
        const int S = 1 + (int)np;
        for (int i = lane; i < S * 10; i += 64) {
            const int sg = i / 10, d = i - sg * 10;
            const uint32_t who = sg == 0 ? best_i : f.pushed[sg - 1].idx;
            const uint32_t *r6 = f.rec6 + (size_t)who * PYA_REC_WORDS;
            const uint32_t cum = (r6[d >> 1] >> ((d & 1) * 16)) & 0xffffu;
            f.sc[i] = b.lut[lut_row(nfrag) + (uint32_t)d * (nfrag + 1) + cum];
        }
        wave_lds_sync();
        if (lane < (int)np) {
            const PushedEntry pe = f.pushed[lane];
            const uint64_t gone = best_bits & ~pe.bits, came = pe.bits & ~best_bits;
            const int a = __popcll(best_bits & (gone - 1));
            atomicOr(&f.site_alt[a], 1ull << nth_set_bit(res.site_mask, __builtin_ctzll(came)));
            f.c_site[lane] = (uint32_t)a;
            float best = 0.f;                               /* depth of the largest score gap (Ascore.cpp:164-172) */
            int depth = 0;
            for (int d = 0; d < PYA_NTOP; d++) {
                const float diff = f.sc[d] - f.sc[(lane + 1) * 10 + d];
                if (diff > best) {
                    best = diff;
                    depth = d;
                }
            }
            f.c_depth[lane] = depth;
        }
        for (int i = lane; i < (int)np * 2; i += 64) {
            f.c_tr[i] = 0;
            f.c_cnt[i] = 0;
        }
        wave_lds_sync();
        /* ---- site-determining ions from the recorded lists (cpp/ModifiedPeptide.cpp:259-320): an
         * ion survives when the other signature's list has no ion within mz_error of it.  The lists
         * are ascending and position-indexed, so the candidates sit at the ion's own index and its
         * neighbours; with at most one partner per ion the reference's greedy walk cancels exactly
         * the partnered pairs (localize_core.hip.h), an ion with two partners hands the PSM over. ---- */
        const float err = cfg->mz_error;
        const int ndir = BOTH ? 2 : 1;
        const int items = (int)np * ndir * 2 * Lm1;
        const FastDiv divL = fastdiv_make((uint32_t)(Lm1 > 0 ? Lm1 : 1));
        int p2 = 1;
        while (p2 <= Lm1) p2 <<= 1;                          /* the search covers indices 0 .. Lm1 */
        bool odd = false;
        if (!(b.debug & 1))
        for (int base = 0; base < items; base += 64) {
            const int e = base + lane;
            if (e < items) {
                const uint32_t ts = fastdiv((uint32_t)e, divL);
                const int i = e - (int)ts * Lm1;
                const int side = (int)ts & 1;
                const int d = BOTH ? ((int)ts >> 1) & 1 : 0;
                const int c = BOTH ? (int)ts >> 2 : (int)ts >> 1;
                const int col_best = d * N + (int)best_i, col_comp = d * N + (int)f.pushed[c].idx;
                const float *mine = f.mzl + (side ? col_comp : col_best);
                const float *other = f.mzl + (side ? col_best : col_comp);
                const float me = mine[(size_t)i * stride];
                float df[4];
                bool ok[4], sk[4];
#pragma unroll
                for (int uu = 0; uu < 4; uu++) {
                    const int q = i - 1 + uu;
                    ok[uu] = q >= 0 && q < Lm1;
                    const float o = ok[uu] ? other[(size_t)q * stride] : (q < 0 ? -__builtin_huge_valf() : __builtin_huge_valf());
                    df[uu] = side ? (o - me) : (me - o);   /* always (winner's ion) - (competitor's ion) */
                    sk[uu] = side ? (df[uu] <= -err) : (df[uu] >= err);
                }
                const int w1 = (ok[1] && __builtin_fabsf(df[1]) < err) ? 1 : 0;
                const int w2 = (ok[2] && __builtin_fabsf(df[2]) < err) ? 1 : 0;
                const int w3 = (ok[3] && __builtin_fabsf(df[3]) < err) ? 1 : 0;
                int cnt = -1;
                if (sk[0] && !sk[1]) cnt = w1 + w2;          /* first candidate = index i     */
                else if (sk[1] && !sk[2]) cnt = w2 + w3;     /* first candidate = index i + 1 */
                if (cnt < 0) {
                    /* the first candidate is further away (ions between the two moved sites): binary
                     * search for the first ion of the other list that is not skipped */
                    int j = 0;
                    for (int step = p2 >> 1; step > 0; step >>= 1) {
                        const int probe = j + step;
                        const float o = probe - 1 < Lm1 ? other[(size_t)(probe - 1) * stride] : __builtin_huge_valf();
                        const float dd = side ? (o - me) : (me - o);
                        if (side ? (dd <= -err) : (dd >= err)) j = probe;
                    }
                    cnt = 0;
                    for (int q = j; q < j + 2 && q < Lm1; q++) {
                        const float o = other[(size_t)q * stride];
                        const float dd = side ? (o - me) : (me - o);
                        cnt += (__builtin_fabsf(dd) < err) ? 1 : 0;
                    }
                }
                if (cnt > 1) {
                    odd = true;                             /* two partners: the serial walk decides */
                } else if (cnt == 0) {
                    atomicAdd(&f.c_tr[c * 2 + side], 1u);
                    if ((int)f.rkl[(size_t)i * stride + (side ? col_comp : col_best)] <= f.c_depth[c])
                        atomicAdd(&f.c_cnt[c * 2 + side], 1u);
                }
            }
        }
        if (__any(odd)) declined = true;
        wave_lds_sync();
        if (!declined) {
            /* ---- Ascores (cpp/Ascore.cpp:200-209, :239-251, :305-313) ---- */
            float asc_l = 0.f;
            if (lane < (int)np) {
                const uint32_t tr0 = f.c_tr[lane * 2], tr1 = f.c_tr[lane * 2 + 1];
                const uint32_t n0 = f.c_cnt[lane * 2], n1 = f.c_cnt[lane * 2 + 1];
                const uint32_t depth = (uint32_t)f.c_depth[lane];
                if (tr0 > b.lut_n_max || tr1 > b.lut_n_max) {
                    fail = 1;
                } else {
                    const float sc0 = b.lut[lut_row(tr0) + depth * (tr0 + 1) + n0];
                    const float sc1 = b.lut[lut_row(tr1) + depth * (tr1 + 1) + n1];
                    asc_l = sc0 - sc1;
                }
            }
            for (int c = 0; c < (int)np; c++) {
                const float asc = __shfl(asc_l, c, 64);
                if (lane == (int)f.c_site[c]) my_asc = asc < my_asc ? asc : my_asc;
            }
        }
    }
    if (declined) {
        /* ---- hand-over: leave what score_signatures would have left ---- */
        if (sig_lane) {
            b.ws[s0 + lane] = my_ws;
            if (b.rec) {
                const uint32_t *r6 = f.rec6 + (size_t)lane * PYA_REC_WORDS;
                uint32_t *dst = b.rec + (s0 + lane) * PYA_REC_WORDS;
#pragma unroll
                for (int d = 0; d < PYA_REC_WORDS; d++) dst[d] = r6[d];
            }
        }
        ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)f.grid)[lane];
        return true;
    }
    if (lane < k && f.site_tie[lane]) my_asc = 0.f < my_asc ? 0.f : my_asc;
    if (lane < k && lane < (int)max_k) {
        out_asc[lane] = my_asc;
        out_alt[lane] = f.site_alt[lane];
    }
    const bool any_fail = __any(fail != 0);
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (any_fail) b.status[psm] = PYA_ST_LUT_RANGE;
    }
    return false;
}

#endif
