/* walk_core.hip.h -- the per-(signature, direction) fragment walker of score_signatures.hip.
 * See score_signatures.hip for the notes. */
#ifndef PYA_WALK_CORE_H
#define PYA_WALK_CORE_H
#include "device_common.hip.h"

/* Per-wave LDS the walkers work on:
 *   resd [64] {m0, m1}: unmodified / modified mass of every residue, so a lane reads the residue
 *        that enters its fragment (index step, or L-1-step when it travels from the C-terminus)
 *        with one ds_read_b64 instead of four v_readlane and selects;
 *   resn [64]: the residues' neutral-loss classes (only staged when the scorer has neutral losses);
 *   cnt  [PYA_NTOP/2][64]: rank histogram, one column per lane, two 16-bit counters per word,
 *        bumped with ds_add_u32 -- a handful of VALU instructions per fragment where a register
 *        histogram of packed fields took ~20.  The score kernel's speed follows its occupancy,
 *        which LDS decides, hence the packing. */
struct WalkEnv {
    const DevConfig *cfg;
    const uint16_t *nl_present;
    const float *nl_uniq;
    const float2 *resd;
    const uint8_t *resn;
    uint32_t *cnt;
    int n_nl, L, zmax;
};

DEV void hist_clear(const WalkEnv &e) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 0; d < PYA_NTOP / 2; d++) e.cnt[d * 64 + lane] = 0u;
}
/* (branch-free: a lane without a hit adds zero, wherever its row index points -- rows 5..7 of a miss
 * lie in whatever follows the histogram inside the workgroup's LDS, and adding zero changes nothing) */
DEV void hist_bump(uint32_t *col, bool active, int rank) {
    const bool hit = active && rank < PYA_NTOP;
    uint32_t row = (uint32_t)rank >> 1;
    asm("" : "+v"(row));                                   /* (keeps shift + shift-add: 2 instructions, not 3) */
    atomicAdd(col + row * 64u, hit ? 1u << (((uint32_t)rank << 4) & 31u) : 0u);
}

/* count of rank d in column `lane` */
DEV uint32_t hist_count(const uint32_t *cnt, int lane, int d) {
    return (cnt[(d >> 1) * 64 + lane] >> ((d & 1) * 16)) & 0xffffu;
}

/* Cumulative counts (Ascore.cpp:115-118: counts[d] = fragments matched at rank <= d) are kept in three registers per
 * walker, a byte per depth (a PSM of this kernel has at most 255 fragments per site assignment): a fragment that
 * matched rank r adds 1 to the bytes d >= r, which is one 16-byte LDS read of the entry below and three adds of the
 * cheap class -- where bumping a histogram column in LDS took four instructions of the expensive class and an LDS
 * atomic, and the scores then had to sum the histogram up.  Entry 15 (no match) is all zeros. */
DEV uint4 fused_cum_entry(uint32_t r) {
    const uint32_t full = 0x01010101u;
    uint4 e;
    e.x = r < 4u ? full << (8u * r) : 0u;
    e.y = r <= 4u ? full : (r < 8u ? full << (8u * (r - 4u)) : 0u);
    e.z = r <= 8u ? 0x0101u : (r == 9u ? 0x0100u : 0u);
    e.w = 0u;
    if (r >= (uint32_t)PYA_NTOP) e.x = e.y = e.z = 0u;
    return e;
}
struct CumCounts {
    uint32_t a, b, c;        /* depths 0-3 | 4-7 | 8-9 */
    DEV void add(const uint4 &e) {
        a += e.x;
        b += e.y;
        c += e.z;
    }
    DEV uint32_t at(int d) const { return ((d < 4 ? a : (d < 8 ? b : c)) >> ((d & 3) * 8)) & 0xffu; }
};

/* stages resd / resn from the one-residue-per-lane registers (caller syncs afterwards) */
DEV void stage_residues(const Residues &res, float2 *resd, uint8_t *resn) {
    const int i = lane_id();
    if (i < res.L) {
        resd[i] = make_float2(res.m0, res.m1);
        if (resn) resn[i] = (uint8_t)res.nl;
    }
}

/* resumable state of a walker: float32 running sum and neutral-loss stack state */
struct WalkState {
    float running;
    uint32_t nl_state;
};

/* General walker: one direction per lane (`dir` may differ between lanes).  Bumps the lane's
 * histogram column and adds to nfrag. */
DEV void walk_range(const WalkEnv &e, const PeakTable &tab, uint64_t resmask, int dir, bool active,
                    int step_begin, int step_end, WalkState &st, uint32_t &nfrag) {
    const DevConfig *cfg = e.cfg;
    const int n_f = cfg->n_fwd, n_b = cfg->n_types - cfg->n_fwd;
    const int my_types = dir == 0 ? n_f : n_b;
    const int t_base = dir == 0 ? 0 : n_f;
    const bool any_f = __any(active && dir == 0), any_b = __any(active && dir == 1);
    const int t_max = (any_f && any_b) ? (n_f > n_b ? n_f : n_b) : (any_f ? n_f : n_b);
    const uint64_t types64 = load_types64(cfg);
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - e.L)) : resmask;
    uint32_t *col = e.cnt + lane_id();
    float running = st.running;                            /* 0 at the start: 0 + r == r exactly */
    uint32_t nl_state = st.nl_state;
    for (int step = step_begin; step < step_end; step++) {
        const int ri = dir ? e.L - 1 - step : step;
        const float2 mm = e.resd[ri];
        const bool mod = (tmask >> step) & 1ull;
        const float r = mod ? mm.y : mm.x;
        running = r + running;                                           /* ModifiedPeptide.cpp:385-389 */
        uint32_t pm = active ? 1u : 0u;
        if (e.n_nl) {
            const uint32_t nlp = e.resn[ri];
            const uint32_t cls = mod ? (nlp >> 4) : (nlp & 15u);
            if (cls) nl_state = nl_bump(nl_state, cls);
            pm = active ? (uint32_t)e.nl_present[nl_state & 255u] : 0u;
        }
        while (__any(pm != 0)) {
            const bool on = pm != 0;
            const int v = on ? __builtin_ctz(pm) : 0;
            pm &= pm - 1;
            const float x = running - (e.n_nl ? e.nl_uniq[v] : 0.f);   /* float subtract (:572) */
            const double xd = (double)x;
            for (int t = 0; t < t_max; t++) {
                const bool on_t = on && t < my_types;
                double A, B;
                type_constants(type_at(types64, t_base + (t < my_types ? t : 0)), &A, &B);
                const double m = (xd + A) - B;
                for (int z = 1; z <= e.zmax; z++) {
                    const float f = charge_mz(m, z);
                    if (on_t) {
                        hist_bump(col, true, match_rank_lds(tab, f));
                        nfrag++;
                    }
                }
            }
        }
    }
    st.running = running;
    st.nl_state = nl_state;
}

/* Fast path for the common scorer settings -- no neutral losses, at most one ion type per
 * direction (BASELINE cfg1/2/3/5): straight-line code per residue step plus a short loop over the
 * higher charges; the number of fragments is steps x charges, so the caller counts them. */
DEV void walk_simple_range(const WalkEnv &e, const PeakTable &tab, uint64_t resmask, int dir, bool active,
                           int step_begin, int step_end, WalkState &st) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    /* per-lane ion-type constants (the type letters are wave-uniform scalars) */
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
    /* bit t = "the t-th residue in THIS lane's travel direction is modified" */
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
    /* the lane's residue pointer moves one entry per step, up or down */
    const float2 *rp = e.resd + (dir ? L - 1 - step_begin : step_begin);
    const int stride = dir ? -1 : 1;
    uint32_t *col = e.cnt + lane_id();
    float running = st.running;                            /* 0 at the start: 0 + r == r exactly */
    for (int step = step_begin; step < step_end; step++, rp += stride) {
        const float2 mm = *rp;
        const uint32_t word = step < 32 ? tlo : thi;       /* wave-uniform choice */
        const bool mod = (word >> (step & 31)) & 1u;
        const float r = mod ? mm.y : mm.x;
        running = r + running;                             /* ModifiedPeptide.cpp:385-389 */
        const double m = ((double)running + A) - B;
        const int rk = match_rank_lds(tab, (float)(m + 1.007825));
        hist_bump(col, active, rk);
        for (int z = 2; z <= e.zmax; z++) hist_bump(col, active, match_rank_lds(tab, charge_mz(m, z)));
    }
    st.running = running;
}

/* "Is the residue of this step modified" for a run of steps, one VALU instruction per step: the
 * travel-order mask is kept MSB-first, and doubling the word shifts it by one step with the bit
 * that falls out as the carry. */
struct StepBits {
    uint32_t w;
    DEV bool next() {
        uint32_t nw;
        const bool c = __builtin_add_overflow(w, w, &nw);
        w = nw;
        return c;
    }
};
/* MSB-first mask of the steps from `begin` on: bit 63 = step `begin` (tmask bit s = step s) */
DEV uint64_t msb_first_from(uint64_t tmask, int begin) { return __brevll(tmask) << begin; }

/* These kernels are bound by the VALU issue rate, so the walkers below count
 * vector instructions: the ion-type offsets that are zero (b: both, y and c: the second) are not
 * added -- x + 0.0 and x - 0.0 are x -- which the two instantiations of the loop know at compile time. */
template <bool BY>
DEV void walk_both_steps(const PeakTable &tab, uint32_t *col, bool active, const float2 *&rp0, const float2 *&rp1,
                         StepBits &b0, StepBits &b1, float &run0, float &run1, double A0, double B0, double A1, double B1,
                         int count) {
    for (int i = 0; i < count; i++, rp0++, rp1--) {
        const float2 m0 = *rp0, m1 = *rp1;
        const float r0 = b0.next() ? m0.y : m0.x, r1 = b1.next() ? m1.y : m1.x;
        run0 = r0 + run0;                                    /* ModifiedPeptide.cpp:385-389 */
        run1 = r1 + run1;
        float f0, f1;
        if (BY) {
            f0 = (float)((double)run0 + 1.007825);
            f1 = (float)(((double)run1 + A1) + 1.007825);
        } else {
            f0 = (float)((((double)run0 + A0) - B0) + 1.007825);
            f1 = (float)((((double)run1 + A1) - B1) + 1.007825);
        }
        const Look k0 = look4(tab, f0), k1 = look4(tab, f1);
        int rk0 = k0.best, rk1 = k1.best;
        if (k0.more()) rk0 = look_rest(tab, k0);
        if (k1.more()) rk1 = look_rest(tab, k1);
        hist_bump(col, active, rk0);
        hist_bump(col, active, rk1);
    }
}
template <bool BY>
DEV void walk_one_steps(const PeakTable &tab, uint32_t *col, bool active, const float2 *&rp, int stride, StepBits &bits,
                        float &run, double A, double B, int count) {
    for (int i = 0; i < count; i++, rp += stride) {
        const float2 mm = *rp;
        const float r = bits.next() ? mm.y : mm.x;
        run = r + run;
        const float f = BY ? (float)(((double)run + A) + 1.007825) : (float)((((double)run + A) - B) + 1.007825);
        const Look k = look4(tab, f);
        int rk = k.best;
        if (k.more()) rk = look_rest(tab, k);
        hist_bump(col, active, rk);
    }
}

/* Both directions of one signature in one loop (charge 1, mz_error <= 0.49): the forward and the
 * backward walker of a lane are independent chains, so every iteration has two lookups in flight.
 * Steps [begin_d, end_d) of direction d; the steps the two ranges have in common run in the paired
 * loop, the rest of the longer one on its own; the mask words change after 32 steps. */
DEV void walk_simple_both(const WalkEnv &e, const PeakTable &tab, uint64_t resmask, bool active, int begin0, int end0,
                          WalkState &st0, int begin1, int end1, WalkState &st1) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double A0 = 0., B0 = 0., A1 = 0., B1 = 0.;
    type_constants(cfg->types[0], &A0, &B0);
    type_constants(cfg->types[cfg->n_fwd], &A1, &B1);
    const bool by = A0 == 0. && B0 == 0. && B1 == 0.;      /* b with y (or c: A1 is whatever it is) */
    const uint64_t M0 = msb_first_from(resmask, begin0), M1 = msb_first_from(__brevll(resmask) >> (64 - L), begin1);
    const float2 *rp0 = e.resd + begin0, *rp1 = e.resd + (L - 1 - begin1);
    uint32_t *col = e.cnt + lane_id();
    float run0 = st0.running, run1 = st1.running;
    const int n0 = end0 - begin0, n1 = end1 - begin1, both = n0 < n1 ? n0 : n1;
    for (int seg = 0; seg < 2; seg++) {                      /* steps 0..31, 32..63 from `begin` */
        StepBits b0 = {seg ? (uint32_t)M0 : (uint32_t)(M0 >> 32)}, b1 = {seg ? (uint32_t)M1 : (uint32_t)(M1 >> 32)};
        const int lo = seg * 32, hi = lo + 32;
        int c = (both < hi ? both : hi) - lo;               /* paired steps of this segment */
        if (c < 0) c = 0;
        if (by) walk_both_steps<true>(tab, col, active, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, c);
        else walk_both_steps<false>(tab, col, active, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, c);
        const int from = lo + c;                            /* first unpaired step of this segment */
        int t0 = (n0 < hi ? n0 : hi) - from, t1 = (n1 < hi ? n1 : hi) - from;
        if (t0 > 0) {
            if (by) walk_one_steps<true>(tab, col, active, rp0, 1, b0, run0, A0, B0, t0);
            else walk_one_steps<false>(tab, col, active, rp0, 1, b0, run0, A0, B0, t0);
        }
        if (t1 > 0) {
            if (by) walk_one_steps<true>(tab, col, active, rp1, -1, b1, run1, A1, B1, t1);
            else walk_one_steps<false>(tab, col, active, rp1, -1, b1, run1, A1, B1, t1);
        }
    }
    st0.running = run0;
    st1.running = run1;
}

/* The simple walkers once more for charge 1 and at most 255 fragments per site assignment, with the cumulative rank
 * counts in three registers per lane (CumCounts: one 16-byte read of `lut` and three adds per fragment) instead of a
 * histogram column in LDS bumped with an atomic -- score_big's way since r04: no histogram to clear, to sum up, or to
 * make room for (10 KB of its workgroup's LDS).  Lanes without work run along; nobody reads their counts. */
DEV void walk_cum_range(const WalkEnv &e, const PeakTable &tab, const uint4 *lut, uint64_t resmask, int dir, int step_begin,
                        int step_end, float &running_io, CumCounts &cum) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
    const float2 *rp = e.resd + (dir ? L - 1 - step_begin : step_begin);
    const int stride = dir ? -1 : 1;
    float running = running_io;
    for (int step = step_begin; step < step_end; step++, rp += stride) {
        const float2 mm = *rp;
        const uint32_t word = step < 32 ? tlo : thi;
        const bool mod = (word >> (step & 31)) & 1u;
        running = (mod ? mm.y : mm.x) + running;           /* ModifiedPeptide.cpp:385-389 */
        const double m = ((double)running + A) - B;
        const Look k = look4(tab, (float)(m + 1.007825));
        int rk = k.best;
        if (k.more()) rk = look_rest(tab, k);
        cum.add(lut[rk]);
    }
    running_io = running;
}
template <bool BY>
DEV void walk_cum_both_steps(const PeakTable &tab, const uint4 *lut, CumCounts &cum, const float2 *&rp0, const float2 *&rp1,
                             StepBits &b0, StepBits &b1, float &run0, float &run1, double A0, double B0, double A1, double B1,
                             int count) {
    for (int i = 0; i < count; i++, rp0++, rp1--) {
        const float2 m0 = *rp0, m1 = *rp1;
        const float r0 = b0.next() ? m0.y : m0.x, r1 = b1.next() ? m1.y : m1.x;
        run0 = r0 + run0;                                    /* ModifiedPeptide.cpp:385-389 */
        run1 = r1 + run1;
        float f0, f1;
        if (BY) {
            f0 = (float)((double)run0 + 1.007825);
            f1 = (float)(((double)run1 + A1) + 1.007825);
        } else {
            f0 = (float)((((double)run0 + A0) - B0) + 1.007825);
            f1 = (float)((((double)run1 + A1) - B1) + 1.007825);
        }
        const Look k0 = look4(tab, f0), k1 = look4(tab, f1);
        int rk0 = k0.best, rk1 = k1.best;
        if (k0.more()) rk0 = look_rest(tab, k0);
        if (k1.more()) rk1 = look_rest(tab, k1);
        cum.add(lut[rk0]);
        cum.add(lut[rk1]);
    }
}
template <bool BY>
DEV void walk_cum_one_steps(const PeakTable &tab, const uint4 *lut, CumCounts &cum, const float2 *&rp, int stride, StepBits &bits,
                            float &run, double A, double B, int count) {
    for (int i = 0; i < count; i++, rp += stride) {
        const float2 mm = *rp;
        const float r = bits.next() ? mm.y : mm.x;
        run = r + run;
        const float f = BY ? (float)(((double)run + A) + 1.007825) : (float)((((double)run + A) - B) + 1.007825);
        const Look k = look4(tab, f);
        int rk = k.best;
        if (k.more()) rk = look_rest(tab, k);
        cum.add(lut[rk]);
    }
}
/* walk_simple_both with register counts: steps [begin_d, end_d) of direction d, running sums in and out */
DEV void walk_cum_both(const WalkEnv &e, const PeakTable &tab, const uint4 *lut, uint64_t resmask, int begin0, int end0, float &run0_io,
                       int begin1, int end1, float &run1_io, CumCounts &cum) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double A0 = 0., B0 = 0., A1 = 0., B1 = 0.;
    type_constants(cfg->types[0], &A0, &B0);
    type_constants(cfg->types[cfg->n_fwd], &A1, &B1);
    const bool by = A0 == 0. && B0 == 0. && B1 == 0.;
    const uint64_t M0 = msb_first_from(resmask, begin0), M1 = msb_first_from(__brevll(resmask) >> (64 - L), begin1);
    const float2 *rp0 = e.resd + begin0, *rp1 = e.resd + (L - 1 - begin1);
    float run0 = run0_io, run1 = run1_io;
    const int n0 = end0 - begin0, n1 = end1 - begin1, both = n0 < n1 ? n0 : n1;
    for (int seg = 0; seg < 2; seg++) {
        StepBits b0 = {seg ? (uint32_t)M0 : (uint32_t)(M0 >> 32)}, b1 = {seg ? (uint32_t)M1 : (uint32_t)(M1 >> 32)};
        const int lo = seg * 32, hi = lo + 32;
        int c = (both < hi ? both : hi) - lo;
        if (c < 0) c = 0;
        if (by) walk_cum_both_steps<true>(tab, lut, cum, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, c);
        else walk_cum_both_steps<false>(tab, lut, cum, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, c);
        const int from = lo + c;
        int t0 = (n0 < hi ? n0 : hi) - from, t1 = (n1 < hi ? n1 : hi) - from;
        if (t0 > 0) {
            if (by) walk_cum_one_steps<true>(tab, lut, cum, rp0, 1, b0, run0, A0, B0, t0);
            else walk_cum_one_steps<false>(tab, lut, cum, rp0, 1, b0, run0, A0, B0, t0);
        }
        if (t1 > 0) {
            if (by) walk_cum_one_steps<true>(tab, lut, cum, rp1, -1, b1, run1, A1, B1, t1);
            else walk_cum_one_steps<false>(tab, lut, cum, rp1, -1, b1, run1, A1, B1, t1);
        }
    }
    run0_io = run0;
    run1_io = run1;
}

/* ---------------------------------------------------------------------------------------------------------------
 * The count-node table (plain settings: no neutral losses, fragment charge 1, one ion type per direction).
 *
 * A fragment's m/z depends on the site assignment only through HOW MANY of the residues it contains are modified --
 * up to rounding: the walkers' float32 running sums add the same masses in the same order, except that different
 * residues carry the modification.  The set of running sums that the site assignments with j modified residues among
 * the first s + 1 (in travel order) can have at step s is enclosed EXACTLY by a recurrence over (s, j): float32
 * addition is monotone (x <= y implies fl(r + x) <= fl(r + y)), so
 *     lo(s, j) = min( fl(m0_s + lo(s-1, j)),  fl(m1_s + lo(s-1, j-1)) if residue s is modifiable ),   hi(s, j) alike with max,
 * lo(-1, 0) = hi(-1, 0) = 0, are the smallest and the largest sum any chain through the node has (cnt_envelopes; the
 * interval is a few float32 ulps wide).  The m/z and the window ends are monotone functions of the sum as well
 * (double additions of constants, a narrowing, a float32 subtraction / addition of the tolerance), so with
 *     a0 = f32(f(lo) - err), a = f32(f(hi) - err), b0 = f32(f(lo) + err), b1 = f32(f(hi) + err)
 * every walker through the node has its window's lower end in [a0, a] and its upper end in [b0, b1], and ONE scan of
 * the peak table decides the fragment for all of them (the reference's test is lower < peak < upper, strict,
 * cpp/ModifiedPeptide.cpp:133-135):
 *     a peak with a < p < b0 matches for every walker, a peak with p <= a0 or p >= b1 for none;
 *     the node's rank is the lowest rank among the former; a peak in between -- within a few ulps of a window end --
 *     with a LOWER rank could change some walker's answer: the node is then MARKED and a walker through it looks its
 *     fragment up itself, with its own sum (exact by construction).
 * Entry: rank (0 .. 9, PYA_NO_MATCH = 15) | 0x80 when marked.  C(15,5) = 3003 site assignments x 58 fragments read
 * 2 x 29 x 6 = 348 node lookups instead of making 64 000 of their own (score_big's two-level prefix tree included).
 * Table layout: entry of (direction d, step s, j modified so far) at t[((d * pos_cap + s) * kc) + j].
 *
 * The counts of a site assignment are then a sum over the nodes on its path, and j is constant between two of its
 * modified residues: with P(d, j, s) = the sum of the nodes' increments over the steps below s (cnt_prefix_sums:
 * cumulative counts as packed bytes, marked nodes counted in a fourth word) a direction's counts are k + 1 differences
 * P(d, i, e_{i+1}) - P(d, i, e_i) over the steps e_i at which its modified residues enter the fragment -- O(k) table
 * reads per site assignment instead of a walk over L - 1 residues (cnt_eval).  A site assignment whose path crosses a
 * marked node (the fourth word says so) is walked (walk_cnt_both).
 * PYA_DEBUG 0x8000: no table (every walker looks every fragment up itself); 0x40000000: every node marked (the table is
 * read, then every walker looks up itself): the three must agree. */
#define CNT_MARK 0x80u

/* lo / hi of every node (see above), by one wavefront: lane = d * 32 + j (k + 1 <= 32), written to
 * env[(d * (k + 1) + j) * pos_cap + s] as float2 {lo, hi}; an unreachable node gets lo > hi.  The residues come from the
 * wavefront's registers (one residue per lane: a v_readlane per step, the step being wave-uniform) and the neighbour's
 * state by a whole-wave DPP shift -- the chain of L - 1 dependent steps makes no LDS round trip. */
DEV float cnt_lane_prev_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
DEV void cnt_envelopes(const Residues &res, int k, uint32_t pos_cap, float2 *env) {
    const int lane = lane_id();
    const int d = lane >> 5, j = lane & 31, L = res.L;
    const float inf = __builtin_huge_valf();
    /* an unreachable node is lo = +inf, hi = -inf: the additions keep it so (the masses are finite), so no flag travels.
     * Lane j = 0 of either half has no neighbour below: the DPP shift hands it +inf / -inf (lanes 31 and 63 hold j = 31, never
     * reachable: k + 1 <= 31). */
    float lo = j == 0 ? 0.f : inf, hi = j == 0 ? 0.f : -inf;
    float2 *out = env + (size_t)(d * (k + 1) + j) * pos_cap;
    const bool keep = j <= k;
    for (int s = 0; s + 1 < L; s++) {
        const int rf = s, rb = L - 1 - s;
        const float m0f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), rf));
        const float m1f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), rf));
        const float m0b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), rb));
        const float m1b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), rb));
        /* a residue that cannot be modified offers no second way in: its "modified mass" is +inf for lo, -inf for hi */
        const bool site_f = (res.site_mask >> rf) & 1ull, site_b = (res.site_mask >> rb) & 1ull;
        const float m1lf = site_f ? m1f : inf, m1hf = site_f ? m1f : -inf, m1lb = site_b ? m1b : inf, m1hb = site_b ? m1b : -inf;   /* (scalar selects) */
        const float m0 = d ? m0b : m0f, m1l = d ? m1lb : m1lf, m1h = d ? m1hb : m1hf;
        float lo_m = cnt_lane_prev_f32(lo), hi_m = cnt_lane_prev_f32(hi);
        if (j == 0) {
            lo_m = inf;
            hi_m = -inf;
        }
        const float a = m0 + lo, c = m1l + lo_m, e = m0 + hi, g = m1h + hi_m;
        lo = a < c ? a : c;
        hi = e > g ? e : g;
        if (keep) out[s] = make_float2(lo, hi);
    }
}

/* One fragment ion of a node: its smallest and largest m/z over the chains through the node (f_lo <= f_hi) against the
 * staged peak table: the rank every walker finds, | CNT_MARK when a peak between the windows could make walkers differ.
 * (mz_error <= 0.49: no half check) */
#ifndef PYA_CNT_SCAN_LOOP
/* r06: the first four entries from the cell's (even) start as two 16-byte reads and a straight line of selects -- what
 * look4 does for a walker's lookup -- and the scan loop only for a window that reaches past them (rare: a cell holds less than
 * one retained peak on average).  As a per-lane loop every wavefront ran to its longest lane, with the exec-mask
 * bookkeeping of two exits per step.  (a0 <= a and b0 <= b1 -- float subtraction and addition are monotone -- so "inside
 * every walker's window" implies "inside the widest one".) */
DEV uint32_t cnt_entry_f(const PeakTable &t, float f_lo, float f_hi) {
    const float a0 = f_lo - t.err, a = f_hi - t.err, b0 = f_lo + t.err, b1 = f_hi + t.err;
    PeakEntry e4[4];
    int idx = entries4(t, (uint32_t)t.cell[grid_cell(t, a0)], e4);
    int in_best = PYA_NO_MATCH, band_best = PYA_NO_MATCH;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const PeakEntry x = e4[j];
        const int rw = (x.mz > a0 && x.mz < b1) ? (int)x.rank : PYA_NO_MATCH;
        const bool core = x.mz > a && x.mz < b0;
        const int ri = core ? rw : PYA_NO_MATCH, rb = core ? PYA_NO_MATCH : rw;
        in_best = ri < in_best ? ri : in_best;
        band_best = rb < band_best ? rb : band_best;
    }
    if (e4[3].mz < b1) {
        for (;; idx++) {
            const PeakEntry x = t.e[idx];
            if (!(x.mz < b1)) break;
            if (x.mz > a0) {
                const int r = (int)x.rank;
                if (x.mz > a && x.mz < b0) in_best = r < in_best ? r : in_best;
                else band_best = r < band_best ? r : band_best;
            }
        }
    }
    return (uint32_t)in_best | (band_best < in_best ? CNT_MARK : 0u);
}
/* ... when the envelope is a point (f_lo == f_hi: no band, nothing to mark) */
DEV uint32_t cnt_entry_1(const PeakTable &t, float f) {
    const Look k = look4(t, f);
    return (uint32_t)(k.more() ? look_rest(t, k) : k.best);
}
#else
DEV uint32_t cnt_entry_f(const PeakTable &t, float f_lo, float f_hi) {
    const float a0 = f_lo - t.err, a = f_hi - t.err, b0 = f_lo + t.err, b1 = f_hi + t.err;
    int in_best = PYA_NO_MATCH, band_best = PYA_NO_MATCH;
    for (int idx = (int)t.cell[grid_cell(t, a0)];; idx++) {      /* every peak > a0 has index >= that; the sentinels end the scan */
        const PeakEntry x = t.e[idx];
        if (!(x.mz < b1)) break;
        if (x.mz > a0) {
            const int r = (int)x.rank;
            if (x.mz > a && x.mz < b0) in_best = r < in_best ? r : in_best;
            else band_best = r < band_best ? r : band_best;
        }
    }
    return (uint32_t)in_best | (band_best < in_best ? CNT_MARK : 0u);
}
/* ... when the envelope is a point (f_lo == f_hi: no band, nothing to mark) */
DEV uint32_t cnt_entry_1(const PeakTable &t, float f) {
    const float a = f - t.err, b = f + t.err;
    int best = PYA_NO_MATCH;
    for (int idx = (int)t.cell[grid_cell(t, a)];; idx++) {
        const PeakEntry x = t.e[idx];
        if (!(x.mz < b)) break;
        if (x.mz > a) best = (int)x.rank < best ? (int)x.rank : best;
    }
    return (uint32_t)best;
}
#endif
/* One node of the plain settings (one ion: charge 1): the envelope [lo, hi] of its running sums (A, B: the ion type's offsets) */
DEV uint32_t cnt_table_entry(const PeakTable &t, float lo, float hi, double A, double B) {
    if (!(lo <= hi)) return (uint32_t)PYA_NO_MATCH;           /* unreachable: never read */
    return cnt_entry_f(t, (float)((((double)lo + A) - B) + 1.007825), (float)((((double)hi + A) - B) + 1.007825));
}

/* P(d, j, s) for s = 0 .. L - 1 at P[(d * (k + 1) + j) * L + s]: x, y, z = the packed cumulative-count increments of the
 * nodes (d, j, t < s) summed (bytes: at most L - 1 per field), w = how many of them are marked.  One wavefront per row
 * (d, j) at a time, one step per lane (L - 1 <= 63), an inclusive scan over the lanes. */
DEV void cnt_prefix_sums(const uint8_t *t, const uint4 *lut, uint32_t pos_cap, uint32_t kc, int L, int k, uint4 *P, int wave, int n_waves) {
    const int lane = lane_id();
    for (int row = wave; row < 2 * (k + 1); row += n_waves) {
        const int d = row / (k + 1), j = row - d * (k + 1);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (lane + 1 < L) {
            const uint32_t ent = t[((size_t)d * pos_cap + lane) * kc + j];
            v = lut[ent & 15u];
            v.w = ent >> 7;
        }
        v.x = wave_incl_scan_u32<false>(v.x);
        v.y = wave_incl_scan_u32<false>(v.y);
        v.z = wave_incl_scan_u32<false>(v.z);
        v.w = wave_incl_scan_u32<false>(v.w);
        uint4 *out = P + (size_t)row * L;
        if (lane == 0) out[0] = make_uint4(0u, 0u, 0u, 0u);
        if (lane + 1 < L) out[lane + 1] = v;
    }
}

/* The counts of a site assignment from the prefix sums.  Direction d, modified residues entering at travel steps
 * e_1 < ... < e_k (e_0 = 0, e_{k+1} = L - 1):  sum_i P(d, i, e_{i+1}) - P(d, i, e_i)  =  P(d, k, L - 1) + sum_{i=1..k} Q(d, i, e_i),
 * Q(d, i, s) = P(d, i - 1, s) - P(d, i, s)  (P(d, 0, 0) = 0).  The words are added and subtracted as integers mod 2^32: a
 * byte field of a Q may be "negative" and borrow from its neighbour, but the final fields are true counts below 256, so
 * the final word is their packed form whatever happened in between.  The t-th modified SITE of a site assignment (sites
 * in N -> C order, t = 1 .. k) is the t-th modification of the forward direction and the (k + 1 - t)-th of the backward
 * one, and a site's residue fixes both steps, so one table serves both directions:
 *     G(t, site) = Q(0, t, min(pos, L - 1)) + Q(1, k + 1 - t, min(L - 1 - pos, L - 1)),     pos = the site's residue,
 *     counts = C + sum_t G(t, site_t),   C = P(0, k, L - 1) + P(1, k, L - 1)
 * -- k reads of 16 bytes per site assignment.  (A modified last residue enters no fragment: its step is capped at
 * L - 1, where P is the total.)  The fourth word counts the marked nodes on the two paths.
 * cnt_site_table: G[(t - 1) * n_sites + site] for t = 1 .. k, then C at G[k * n_sites]; one thread per entry. */
DEV void cnt_site_table(const uint4 *P, const uint8_t *site_pos, int L, int k, int n_sites, uint4 *G, int tid, int nthreads) {
    const int Lm1 = L - 1;
    for (int i = tid; i <= k * n_sites; i += nthreads) {
        uint4 g;
        if (i == k * n_sites) {
            const uint4 a = P[(size_t)k * L + Lm1], c = P[(size_t)(k + 1 + k) * L + Lm1];
            g = make_uint4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        } else {
            const int t = i / n_sites + 1, site = i - (t - 1) * n_sites;
            const int pos = (int)site_pos[site];
            const int ef = pos < Lm1 ? pos : Lm1, qb = Lm1 - pos, eb = qb < Lm1 ? qb : Lm1;
            const int tb = k + 1 - t;
            const uint4 f0 = P[(size_t)(t - 1) * L + ef], f1 = P[(size_t)t * L + ef];
            const uint4 b0 = P[(size_t)(k + 1 + tb - 1) * L + eb], b1 = P[(size_t)(k + 1 + tb) * L + eb];
            g = make_uint4((f0.x - f1.x) + (b0.x - b1.x), (f0.y - f1.y) + (b0.y - b1.y), (f0.z - f1.z) + (b0.z - b1.z),
                           (f0.w - f1.w) + (b0.w - b1.w));
        }
        G[i] = g;
    }
}
/* the counts of the site assignment `bits` (bit = site, k bits set); *marked = marked nodes on its two paths */
DEV CumCounts cnt_eval_sites(const uint4 *G, uint32_t bits, int k, int n_sites, uint32_t *marked) {
    uint4 acc = G[k * n_sites];
    const uint4 *row = G;
    for (int t = 0; t < k; t++, row += n_sites) {
        const int site = __builtin_ctz(bits);
        bits &= bits - 1u;
        const uint4 g = row[site];
        acc.x += g.x;
        acc.y += g.y;
        acc.z += g.z;
        acc.w += g.w;
    }
    *marked = acc.w;
    CumCounts c = {acc.x, acc.y, acc.z};
    return c;
}

template <bool BY>
DEV void walk_cnt_both_steps(const PeakTable &tab, const uint4 *lut, CumCounts &cum, const float2 *&rp0, const float2 *&rp1,
                             StepBits &b0, StepBits &b1, float &run0, float &run1, double A0, double B0, double A1, double B1,
                             const uint8_t *&row0, const uint8_t *&row1, uint32_t kc, uint32_t &j0, uint32_t &j1, int count) {
    for (int i = 0; i < count; i++, rp0++, rp1--, row0 += kc, row1 += kc) {
        const float2 m0 = *rp0, m1 = *rp1;
        const bool d0 = b0.next(), d1 = b1.next();
        run0 = (d0 ? m0.y : m0.x) + run0;                    /* ModifiedPeptide.cpp:385-389 */
        run1 = (d1 ? m1.y : m1.x) + run1;
        j0 += d0 ? 1u : 0u;
        j1 += d1 ? 1u : 0u;
        const uint32_t e0 = row0[j0], e1 = row1[j1];
        int rk0 = (int)(e0 & 15u), rk1 = (int)(e1 & 15u);
        if ((e0 | e1) & CNT_MARK) {
            if (e0 & CNT_MARK) {
                const float f0 = BY ? (float)((double)run0 + 1.007825) : (float)((((double)run0 + A0) - B0) + 1.007825);
                const Look k0 = look4(tab, f0);
                rk0 = k0.best;
                if (k0.more()) rk0 = look_rest(tab, k0);
            }
            if (e1 & CNT_MARK) {
                const float f1 = BY ? (float)(((double)run1 + A1) + 1.007825) : (float)((((double)run1 + A1) - B1) + 1.007825);
                const Look k1 = look4(tab, f1);
                rk1 = k1.best;
                if (k1.more()) rk1 = look_rest(tab, k1);
            }
        }
        cum.add(lut[rk0]);
        cum.add(lut[rk1]);
    }
}
template <bool BY>
DEV void walk_cnt_one_steps(const PeakTable &tab, const uint4 *lut, CumCounts &cum, const float2 *&rp, int stride, StepBits &bits,
                            float &run, double A, double B, const uint8_t *&row, uint32_t kc, uint32_t &j, int count) {
    for (int i = 0; i < count; i++, rp += stride, row += kc) {
        const float2 mm = *rp;
        const bool d = bits.next();
        run = (d ? mm.y : mm.x) + run;
        j += d ? 1u : 0u;
        const uint32_t ent = row[j];
        int rk = (int)(ent & 15u);
        if (ent & CNT_MARK) {
            const float f = BY ? (float)(((double)run + A) + 1.007825) : (float)((((double)run + A) - B) + 1.007825);
            const Look k = look4(tab, f);
            rk = k.best;
            if (k.more()) rk = look_rest(tab, k);
        }
        cum.add(lut[rk]);
    }
}
/* walk_cum_range reading the count-node table: `row` = the table row of (dir, step_begin), `j` = modified residues
 * before step_begin (both advance) */
DEV void walk_cnt_range(const WalkEnv &e, const PeakTable &tab, const uint4 *lut, const uint8_t *row, uint32_t kc, uint64_t resmask,
                        int dir, int step_begin, int step_end, float &running_io, uint32_t &j_io, CumCounts &cum) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint64_t M = msb_first_from(tmask, step_begin);
    const float2 *rp = e.resd + (dir ? L - 1 - step_begin : step_begin);
    const int stride = dir ? -1 : 1;
    const int n = step_end - step_begin;
    for (int seg = 0; seg < 2; seg++) {                      /* the mask words change after 32 steps */
        StepBits bits = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
        int c = (n < seg * 32 + 32 ? n : seg * 32 + 32) - seg * 32;
        if (c < 0) c = 0;
        walk_cnt_one_steps<false>(tab, lut, cum, rp, stride, bits, running_io, A, B, row, kc, j_io, c);
    }
}
/* walk_cum_both reading the count-node table: t = the table, rows of direction d start at t + d * pos_cap * kc;
 * j0 / j1 = modified residues before begin0 / begin1 in the respective travel order */
DEV void walk_cnt_both(const WalkEnv &e, const PeakTable &tab, const uint4 *lut, const uint8_t *t, uint32_t pos_cap, uint32_t kc,
                       uint64_t resmask, int begin0, int end0, float &run0_io, uint32_t j0, int begin1, int end1, float &run1_io,
                       uint32_t j1, CumCounts &cum) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    double A0 = 0., B0 = 0., A1 = 0., B1 = 0.;
    type_constants(cfg->types[0], &A0, &B0);
    type_constants(cfg->types[cfg->n_fwd], &A1, &B1);
    const bool by = A0 == 0. && B0 == 0. && B1 == 0.;
    const uint64_t M0 = msb_first_from(resmask, begin0), M1 = msb_first_from(__brevll(resmask) >> (64 - L), begin1);
    const float2 *rp0 = e.resd + begin0, *rp1 = e.resd + (L - 1 - begin1);
    const uint8_t *row0 = t + (size_t)begin0 * kc, *row1 = t + ((size_t)pos_cap + (size_t)begin1) * kc;
    float run0 = run0_io, run1 = run1_io;
    const int n0 = end0 - begin0, n1 = end1 - begin1, both = n0 < n1 ? n0 : n1;
    for (int seg = 0; seg < 2; seg++) {
        StepBits b0 = {seg ? (uint32_t)M0 : (uint32_t)(M0 >> 32)}, b1 = {seg ? (uint32_t)M1 : (uint32_t)(M1 >> 32)};
        const int lo = seg * 32, hi = lo + 32;
        int c = (both < hi ? both : hi) - lo;
        if (c < 0) c = 0;
        if (by) walk_cnt_both_steps<true>(tab, lut, cum, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, row0, row1, kc, j0, j1, c);
        else walk_cnt_both_steps<false>(tab, lut, cum, rp0, rp1, b0, b1, run0, run1, A0, B0, A1, B1, row0, row1, kc, j0, j1, c);
        const int from = lo + c;
        int t0 = (n0 < hi ? n0 : hi) - from, t1 = (n1 < hi ? n1 : hi) - from;
        if (t0 > 0) {
            if (by) walk_cnt_one_steps<true>(tab, lut, cum, rp0, 1, b0, run0, A0, B0, row0, kc, j0, t0);
            else walk_cnt_one_steps<false>(tab, lut, cum, rp0, 1, b0, run0, A0, B0, row0, kc, j0, t0);
        }
        if (t1 > 0) {
            if (by) walk_cnt_one_steps<true>(tab, lut, cum, rp1, -1, b1, run1, A1, B1, row1, kc, j1, t1);
            else walk_cnt_one_steps<false>(tab, lut, cum, rp1, -1, b1, run1, A1, B1, row1, kc, j1, t1);
        }
    }
    run0_io = run0;
    run1_io = run1;
}

DEV bool walk_is_simple(const WalkEnv &e) {
    const int n_f = e.cfg->n_fwd, n_b = e.cfg->n_types - e.cfg->n_fwd;
    return e.n_nl == 0 && n_f <= 1 && n_b <= 1;
}

#endif
