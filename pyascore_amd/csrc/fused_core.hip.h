/* fused_core.hip.h -- scoring AND localisation of one PSM by one wavefront, in one pass, for the
 * PSMs that make up most real batches: few site assignments (C(n,k) <= 32, or <= 64 with ion types
 * of one direction only), no neutral losses, one ion type per direction, at most 255 fragments per
 * site assignment (any fragment charge).
 *
 * Replaces, for those PSMs, score_signatures + rank_and_localize (cpp/Ascore.cpp:53-254,
 * cpp/ModifiedPeptide.cpp:126-150, :259-320, :326-609).  The two-kernel route makes every walker
 * throw away what localisation needs again -- the rank of the peak each fragment matched -- and
 * then re-derives it: per-signature prefix tables, fragment lists rebuilt from them, every
 * surviving ion looked up in the peak table in global memory, the per-signature scores, counts and
 * the m/z grid written to HBM by one kernel and read back by the next.  Here every (signature,
 * direction) walker records the rank each of its L-1 fragments matched (one byte per step, in
 * LDS) while it scores; once the winner and its single-move competitors are known, only THEIR
 * fragment m/z are re-derived (a walk without lookups: the float32 running sums) and the
 * site-determining ions become a comparison of those few lists and a count over recorded ranks:
 * no lookup, nothing through HBM but the inputs and the 64-byte result.
 *
 * This kernel's speed follows its occupancy (measured: its time is inversely proportional to the
 * wavefronts resident per CU up to the VALU issue limit -- every wavefront is a chain of dependent
 * LDS and memory round trips; fetching the next PSM's inputs ahead in a persistent wavefront was
 * measured too and lost to the registers and LDS it costs), so its LDS footprint is kept small:
 * ranks instead of m/z lists, competitors localised three at a time, and everything that is dead
 * after the walk (rank histogram, peak table) reused for the bookkeeping that follows.
 *
 * Fragment charges above 1 are in: a list is then one ascending run per charge; an ion looks for
 * partners among its neighbours in the run of its own charge and by binary search in the others.
 * A (competitor, direction) task in which some ion has two partners within mz_error -- common once
 * charge states interleave -- is replayed right here with the reference's serial walk over the two
 * lists, merged by rank computation; only those tasks, one lane each.
 *
 * Exactness is that of the two kernels (same walker, same window test, same std::sort emulation,
 * same pairing rule): a PSM that needs a route this body does not have -- a residue mass that is
 * not positive (lists not ascending), introsort running out of depth -- is handed over: its scores,
 * count records and grid are written where score_signatures would have left them and the general
 * localize instantiation redoes it.
 */
#ifndef PYA_FUSED_CORE_H
#define PYA_FUSED_CORE_H
#include "score_core.hip.h"
#include "localize_core.hip.h"

#define FUSED_ROUND 3                /* competitors localised together (plus the winner) */

/* LDS of one wavefront (byte offsets are computed as integers and added to the LDS base: casting
 * pointers through integers to align them hides the address space from the compiler, which then
 * emits FLAT accesses -- slower, and waited for together with every global load in flight):
 *   grid u16[256]
 *   walk region   cum u32[16][4] | peaks PeakEntry[cap + 4]                     (dead after the walk)
 *   post region   (same bytes) sort arrays, pushed competitors, per-site maxima / ties / alternative
 *                 sites, depth scores and counters of a round
 *   kept          resd float2[pos_cap + 1] | rkl u8[ent_cap][stride] | rec u32[n_cap][3] | wsl f32[n_cap]
 *                 | mass f32[32] | flags u32[32] | site_pos u8[pos_cap + 1] | selm f32[(1 + FUSED_ROUND) * ndir][ent_cap]
 *                 (pos_cap = L - 1, ent_cap = (L - 1) x charges: list entries per (signature, direction)) */
struct FusedLds {
    uint16_t *grid;
    uint4 *cum_lut;          /* [16] rank -> increments of the ten cumulative counts, a byte each (see fused_cum_entry) */
    PeakEntry *peaks;
    float2 *resd;
    uint8_t *rkl;
    uint32_t *rec;           /* per signature: ten cumulative counts as bytes (<= 126 fragments) */
    float *wsl;
    float *mass_l;
    uint32_t *flag_l;
    uint8_t *site_pos;       /* [pos_cap + 1] residue of the j-th modifiable one */
    float *selm;
    /* post region */
    unsigned char *sort_raw;   /* sort_carve(sort_raw, N): the std::sort emulation's work area */
    PushedEntry *pushed;
    unsigned long long *site_alt;
    uint32_t *site_max, *site_tie, *n_pushed;
    float *sc;               /* [1 + FUSED_ROUND][10] depth scores: winner, then the round's competitors */
    uint32_t *c_tr, *c_cnt;  /* [FUSED_ROUND * ndir][2] per task (competitor, direction) and side */
    uint32_t *bad_tasks;     /* bit t: task t has an ion with two partners -> replayed serially */
    uint32_t *t_lo, *t_off;  /* [8] per task: first step of its span, items before it (see the pairing) */
    float *srt_v;            /* [FUSED_ROUND * ndir][2][ent_cap] a task's two lists, sorted (only filled for replays) */
    uint8_t *srt_h;          /* same shape: the ion matched a peak of rank <= depth */
    int32_t *c_depth;        /* [FUSED_ROUND] */
    uint32_t *c_site;        /* [FUSED_ROUND] */
};

__host__ __device__ static inline size_t fused_align16(size_t v) { return (v + 15) & ~(size_t)15; }
/* (sort_lds_bytes of localize_core.hip.h, restated for the host side of this header) */
__host__ __device__ static inline size_t fused_sort_bytes(uint32_t n) {
    if (n <= 64) return ((size_t)n * 10 + 15) & ~(size_t)15;
    return (((size_t)n * 6 + 15) & ~(size_t)15) + ((size_t)n / 64 + 2) * 16 + 2 * 128 * 2;
}
/* multi_z: the merged lists of the multi-charge instantiation (srt_v, srt_h).  The charge-1 instantiation
 * has no use for them -- a task with a doubly partnered ion, which its residue-mass condition all but
 * rules out, is handed to the general kernel instead of being replayed here -- and without them the
 * post region is no larger than the walk region it overlays: a fifth more wavefronts per CU on cfg2. */
/* entries of the three per-site arrays: a PSM's modified sites number at most its single-move competitors (k <= k (n - k)
 * for an ambiguous PSM), which push_cap bounds */
__host__ __device__ static inline uint32_t fused_site_cap(uint32_t push_cap) {
    const uint32_t v = (push_cap + 3u) & ~3u;
    return v > 64u ? 64u : (v < 4u ? 4u : v);
}
__host__ __device__ static inline size_t fused_post_bytes(uint32_t n_cap, uint32_t push_cap, uint32_t ent_cap, uint32_t ndir,
                                                          bool multi_z) {
    return fused_align16(fused_sort_bytes(n_cap)) + (size_t)push_cap * 16 + (size_t)fused_site_cap(push_cap) * 16 + 16 +
           (size_t)(1 + FUSED_ROUND) * 40 + (size_t)FUSED_ROUND * (16 * ndir + 4 + 4) + 16 + 64 +
           (multi_z ? fused_align16((size_t)FUSED_ROUND * ndir * 2 * ent_cap * 4) +
                          fused_align16((size_t)FUSED_ROUND * ndir * 2 * ent_cap)
                    : 0);
}
__host__ __device__ static inline size_t fused_walk_bytes(uint32_t cap) {
    return 16 * 16 + ((size_t)cap + PYA_TABLE_PAD) * 8;
}
/* pos_cap = largest L - 1 of the launch, ent_cap = largest (L - 1) x charges: list entries per
 * (signature, direction) */
__host__ __device__ static inline size_t fused_kept_bytes(uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap,
                                                          uint32_t ndir) {
    return fused_align16(((size_t)pos_cap + 1) * 8) + fused_align16((size_t)ent_cap * stride) + (size_t)n_cap * 16 + 256 +
           fused_align16((size_t)pos_cap + 1) + (size_t)(1 + FUSED_ROUND) * ndir * ent_cap * 4;
}
__host__ __device__ static inline size_t fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
                                                         uint32_t ent_cap, uint32_t push_cap, uint32_t ndir, bool multi_z) {
    const size_t walk = fused_walk_bytes(cap), post = fused_post_bytes(n_cap, push_cap, ent_cap, ndir, multi_z);
    return PYA_GRID_CELLS * 2 + fused_align16(walk > post ? walk : post) +
           fused_kept_bytes(n_cap, stride, pos_cap, ent_cap, ndir) + 16;
}

DEV FusedLds fused_carve(unsigned char *raw, uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
                         uint32_t ent_cap, uint32_t push_cap, uint32_t ndir, bool multi_z) {
    FusedLds f;
    f.grid = (uint16_t *)raw;
    size_t o = PYA_GRID_CELLS * 2;
    f.cum_lut = (uint4 *)(raw + o);
    f.peaks = (PeakEntry *)(raw + o + 16 * 16);
    /* post region over the walk region */
    f.sort_raw = raw + o;
    size_t q = o + fused_align16(fused_sort_bytes(n_cap));
    f.pushed = (PushedEntry *)(raw + q);
    q += (size_t)push_cap * sizeof(PushedEntry);
    const size_t sc = fused_site_cap(push_cap);
    f.site_alt = (unsigned long long *)(raw + q);
    q += sc * 8;
    f.site_max = (uint32_t *)(raw + q);
    q += sc * 4;
    f.site_tie = (uint32_t *)(raw + q);
    q += sc * 4;
    f.n_pushed = (uint32_t *)(raw + q);
    q += 16;
    f.sc = (float *)(raw + q);
    q += (size_t)(1 + FUSED_ROUND) * 40;
    f.c_tr = (uint32_t *)(raw + q);
    q += (size_t)FUSED_ROUND * ndir * 8;
    f.c_cnt = (uint32_t *)(raw + q);
    q += (size_t)FUSED_ROUND * ndir * 8;
    f.c_depth = (int32_t *)(raw + q);
    q += FUSED_ROUND * 4;
    f.c_site = (uint32_t *)(raw + q);
    q += FUSED_ROUND * 4;
    f.bad_tasks = (uint32_t *)(raw + q);
    q += 16;
    f.t_lo = (uint32_t *)(raw + q);
    f.t_off = f.t_lo + 8;
    q += 64;
    f.srt_v = (float *)(raw + q);
    q += fused_align16((size_t)FUSED_ROUND * ndir * 2 * ent_cap * 4);
    f.srt_h = (uint8_t *)(raw + q);
    const size_t walk = fused_walk_bytes(cap), post = fused_post_bytes(n_cap, push_cap, ent_cap, ndir, multi_z);
    o += fused_align16(walk > post ? walk : post);
    f.resd = (float2 *)(raw + o);
    o += fused_align16(((size_t)pos_cap + 1) * 8);
    f.rkl = (uint8_t *)(raw + o);
    o += fused_align16((size_t)ent_cap * stride);
    f.rec = (uint32_t *)(raw + o);
    o += (size_t)n_cap * 12;
    f.wsl = (float *)(raw + o);
    o += (size_t)n_cap * 4;
    f.mass_l = (float *)(raw + o);
    f.flag_l = (uint32_t *)(raw + o + 128);
    f.site_pos = (uint8_t *)(raw + o + 256);
    f.selm = (float *)(raw + o + 256 + fused_align16((size_t)pos_cap + 1));
    return f;
}

/* (CumCounts and fused_cum_entry: walk_core.hip.h) */

/* the straight-line walker of walk_core.hip.h that also records the rank every fragment matched in
 * column `w` of rkl (entry step * zmax + z - 1); A, B: the ion-type offsets of the lane's direction.
 * Lanes without a walker run the same code: their counts are never read and they
 * record into a spare column (`w` = stride - 1 for them) -- no per-lane branches in the loop.
 * BZ: no direction subtracts an offset (every ion type but z / Z), so that instruction is left out. */
template <bool BZ>
DEV void walk_record_steps(const float2 *&rp, int rstride, StepBits &bits, float &running, const uint4 *lut, CumCounts &cum,
                           const PeakTable &tab, int zmax, double A, double B, uint8_t *&ro, int stride, int count) {
    for (int i = 0; i < count; i++, rp += rstride) {
        const float2 mm = *rp;
        running = (bits.next() ? mm.y : mm.x) + running;     /* ModifiedPeptide.cpp:385-389 */
        const double m = BZ ? (double)running + A : ((double)running + A) - B;
        for (int z = 1; z <= zmax; z++, ro += stride) {
            const Look k = look4(tab, charge_mz(m, z));
            int rk = k.best;
            if (k.more()) rk = look_rest(tab, k);
            cum.add(lut[rk]);
            *ro = (uint8_t)rk;
        }
    }
}
DEV CumCounts walk_record(const float2 *resd, const uint4 *lut, const PeakTable &tab, int L, int zmax, uint64_t resmask, int dir,
                          double A, double B, bool all_b_zero, uint8_t *rkl, int stride, int w) {
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const float2 *rp = resd + (dir ? L - 1 : 0);
    const int rstride = dir ? -1 : 1;
    uint8_t *ro = rkl + w;
    float running = 0.f;
    CumCounts cum = {0u, 0u, 0u};
    if (tab.half_check) {                                    /* mz_error > 0.49: the lookup with the extra test */
        const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
        for (int step = 0; step + 1 < L; step++, rp += rstride) {
            const float2 mm = *rp;
            const uint32_t word = step < 32 ? tlo : thi;
            const bool mod = (word >> (step & 31)) & 1u;
            running = (mod ? mm.y : mm.x) + running;
            const double m = ((double)running + A) - B;
            for (int z = 1; z <= zmax; z++, ro += stride) {
                const int rk = match_rank_lds(tab, charge_mz(m, z));
                cum.add(lut[rk]);
                *ro = (uint8_t)rk;
            }
        }
        return cum;
    }
    const uint64_t M = msb_first_from(tmask, 0);
    const int n = L - 1;
    for (int seg = 0; seg < 2; seg++) {                      /* the mask words change after 32 steps */
        StepBits bits = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
        int c = (n < seg * 32 + 32 ? n : seg * 32 + 32) - seg * 32;
        if (c < 0) c = 0;
        if (all_b_zero) walk_record_steps<true>(rp, rstride, bits, running, lut, cum, tab, zmax, A, B, ro, stride, c);
        else walk_record_steps<false>(rp, rstride, bits, running, lut, cum, tab, zmax, A, B, ro, stride, c);
    }
    return cum;
}

/* the same walk without lookups: fragment m/z of one signature and direction to out[z-1][step] */
DEV void walk_mz_only(const float2 *resd, int L, int zmax, uint64_t resmask, int dir, double A, double B, float *out) {
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint64_t M = msb_first_from(tmask, 0);
    const float2 *rp = resd + (dir ? L - 1 : 0);
    const int rstride = dir ? -1 : 1;
    const int Lm1 = L - 1;
    float running = 0.f;
    int step = 0;
    for (int seg = 0; seg < 2; seg++) {
        StepBits bits = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
        const int end = Lm1 < seg * 32 + 32 ? Lm1 : seg * 32 + 32;
        for (; step < end; step++, rp += rstride) {
            const float2 mm = *rp;
            running = (bits.next() ? mm.y : mm.x) + running;
            const double m = ((double)running + A) - B;
            out[step] = (float)(m + 1.007825);
            for (int z = 2; z <= zmax; z++) out[(z - 1) * Lm1 + step] = charge_mz(m, z);
        }
    }
}

/* ... and its first half alone: the float32 running sums (charge 1: a second, wave-wide pass turns them into m/z) */
DEV void walk_sums_only(const float2 *resd, int L, uint64_t resmask, int dir, float *out) {
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint64_t M = msb_first_from(tmask, 0);
    const float2 *rp = resd + (dir ? L - 1 : 0);
    const int rstride = dir ? -1 : 1;
    const int Lm1 = L - 1;
    float running = 0.f;
    int step = 0;
    for (int seg = 0; seg < 2; seg++) {
        StepBits bits = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
        const int end = Lm1 < seg * 32 + 32 ? Lm1 : seg * 32 + 32;
        for (; step < end; step++, rp += rstride) {
            const float2 mm = *rp;
            running = (bits.next() ? mm.y : mm.x) + running;
            out[step] = running;
        }
    }
}

/* ions of one ascending run within mz_error of `me` (0, 1, or 2 = "more than one"): binary search
 * for the first ion that is not skipped, then the two candidates from there (localize_core.hip.h) */
DEV int partners_in_run(const float *run, int n, int p2, float me, int side, float err) {
    int j = 0;
    for (int step = p2 >> 1; step > 0; step >>= 1) {
        const int probe = j + step;
        const float o = probe - 1 < n ? run[probe - 1] : __builtin_huge_valf();
        const float dd = side ? (o - me) : (me - o);
        if (side ? (dd <= -err) : (dd >= err)) j = probe;
    }
    int cnt = 0;
    for (int q = j; q < j + 2 && q < n; q++) {
        const float dd = side ? (run[q] - me) : (me - run[q]);
        cnt += (__builtin_fabsf(dd) < err) ? 1 : 0;
    }
    return cnt;
}

/* BOTH: ion types of both directions -- lanes 0..31 walk from the N-terminus, lanes 32..63 from the
 * C-terminus (C(n,k) <= 32); otherwise one direction, one signature per lane (C(n,k) <= 64).
 * Returns true when the PSM was handed over to the general localize instantiation. */
/* ZM: fragment charges above 1 possible (otherwise the charge loops compile away). */
/* The retained table as bin_core has just left it in this wavefront's LDS (float m/z array, byte rank array):
 * a caller that binned the spectrum itself passes it instead of having it read back from the workspace
 * (tiny_batch.hip: pya_one_kernel).  Up to FUSED_LOCAL_CHUNKS x 64 retained peaks. */
#define FUSED_LOCAL_CHUNKS 8
struct LocalTable {
    const float *mz;
    const uint8_t *rank;
    int R, status;
};

/* WIDE (BOTH only): the launch holds PSMs of 33 .. 64 site assignments, which walk their two directions one after the other,
 * a lane per site assignment (an instantiation of its own: the walk of the others -- cfg2's -- stays as it was) */
template <bool BOTH, bool ZM, bool WIDE = false>
DEV bool fused_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t n_cap, uint32_t stride,
                    uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap, const LocalTable *local = nullptr) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const int ndir = BOTH ? 2 : 1;
    const FusedLds f = fused_carve(lds_raw, cap, n_cap, stride, pos_cap, ent_cap, push_cap, (uint32_t)ndir, ZM);
    STAMP_BEGIN();
    STAMP_T(b, 38, false);
    /* a table handed over in LDS sits where this body's own arrays go: into registers before anything is written */
    float lmz[FUSED_LOCAL_CHUNKS];
    uint32_t lrk[FUSED_LOCAL_CHUNKS];
    if (local) {
#pragma unroll
        for (int q = 0; q < FUSED_LOCAL_CHUNKS; q++) {
            const int i = q * 64 + lane_id();
            lmz[q] = i < local->R ? local->mz[i] : __builtin_huge_valf();
            lrk[q] = i < local->R ? (uint32_t)local->rank[i] : (uint32_t)PYA_NO_MATCH;
        }
        wave_lds_sync();
    }
    /* the residue table does not depend on the PSM: on its way before anything else */
    if (lane < 32) {
        f.mass_l[lane] = cfg->res_mass[lane];
        f.flag_l[lane] = cfg->res_modifiable[lane];
    }
    /* everything the PSM's other loads depend on sits in one cache line (common.h: PYA_DESC_WORDS) */
    const uint64_t *dw = b.desc + (size_t)psm * PYA_DESC_WORDS;
    const int64_t p0 = (int64_t)dw[0], pep0 = (int64_t)dw[1], s0 = (int64_t)dw[2], a0 = (int64_t)dw[3];
    const uint64_t w4 = dw[4], w5 = dw[5];
    const int L = (int)(w4 & 0xffffu), n_aux = (int)((w4 >> 16) & 0xffffu), k = (int)((w4 >> 32) & 0xffffu);
    const int zmax = ZM ? (int)(w4 >> 56) : 1;
    const int N = (int)(uint32_t)w5;
    const uint64_t *order = b.order_tab + (uint32_t)(w5 >> 32);
    const int status = local ? local->status : b.status[psm];
    const int R = local ? local->R : (int)b.ret_n[psm];

    const uint32_t max_k = b.max_k;
    float *out_asc = b.ascores + (size_t)psm * max_k;
    uint64_t *out_alt = b.alt_mask + (size_t)psm * max_k;
    for (uint32_t a = lane; a < max_k; a += 64) {
        out_asc[a] = 0.f;
        out_alt[a] = 0ull;
    }
    if (status != PYA_ST_OK) {
        if (lane == 0) {
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return false;
    }
    STAMP_T(b, 39, false);
    /* ---- inputs: letters, retained peaks, the lane's signature, fixed modifications ---- */
    /* BOTH with 33 .. 64 site assignments (r05): a lane per site assignment and the two directions one after the other --
     * twice the walk, still one kernel instead of score_signatures + localize */
    const bool two_pass = BOTH && WIDE && N > 32;
    const int s = (BOTH && !two_pass) ? (lane & 31) : lane;
    const int dir = BOTH ? (two_pass ? 0 : (lane >> 5)) : (cfg->n_fwd > 0 ? 0 : 1);
    const bool active = s < N;
    const uint32_t letter = lane < L ? (uint32_t)b.pep[pep0 + lane] : (uint32_t)'A';
    const uint64_t bits = active ? order[s] : 0ull;
    uint32_t aux_pos = 0;
    float aux_mass = 0.f;
    if (lane < n_aux) {
        aux_pos = b.aux_pos[a0 + lane];
        aux_mass = b.aux_mass[a0 + lane];
    }
    if (local) {
#pragma unroll
        for (int q = 0; q < FUSED_LOCAL_CHUNKS; q++) {
            const int i = q * 64 + lane;
            if (i < R + PYA_TABLE_PAD) {                     /* (past the last peak: the registers hold the sentinels) */
                PeakEntry e;
                e.mz = lmz[q];
                e.rank = lrk[q];
                f.peaks[i] = e;
            }
        }
    } else {
        copy_peak_table(b.ret + p0, R, f.peaks, lane, 64);
    }
    PeakTable tab;
    tab.e = f.peaks;
    tab.g_cell = nullptr;
    tab.g_e = b.ret + p0;
    tab.n = R;
    tab.err = cfg->mz_error;
    tab.half_check = cfg->mz_error > 0.49f;
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;              /* ion-type offsets per direction of travel */
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    wave_lds_sync();
    STAMP_T(b, 48, false);
    /* ---- residues (ModifiedPeptide.cpp:24-79) from the letter and the LDS table ---- */
    float m0, m1;
    uint64_t site_mask;
    {
        const bool in = lane < L;
        const uint32_t li = (letter - 'A') & 31u;
        m0 = f.mass_l[li];
        const bool modifiable = in && ((f.flag_l[li] & 1u) || (cfg->allow_n && lane == 0) || (cfg->allow_c && lane == L - 1));
        m1 = m0 + cfg->mod_mass;
        for (int base = 0; base < n_aux; base += 64) {
            const int n = n_aux - base < 64 ? n_aux - base : 64;
            if (base > 0) {                                  /* (more than 64 fixed modifications) */
                aux_pos = 0;
                aux_mass = 0.f;
                if (lane < n) {
                    aux_pos = b.aux_pos[a0 + base + lane];
                    aux_mass = b.aux_mass[a0 + base + lane];
                }
            }
            for (int j = 0; j < n; j++) {
                const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)aux_pos, j);
                const float am = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(aux_mass), j));
                const int idx = pos > 0 ? (int)pos - 1 : 0;
                if (idx == lane) {
                    m0 += am;
                    m1 += am;
                }
            }
        }
        site_mask = __ballot(modifiable);
        if (in) f.resd[lane] = make_float2(m0, m1);
        if (modifiable) f.site_pos[mask_rank(site_mask)] = (uint8_t)lane;
    }
    STAMP_T(b, 49, false);
    grid_build(&tab, f.grid);
    STAMP_T(b, 50, false);
    if (lane < 16) f.cum_lut[lane] = fused_cum_entry((uint32_t)lane);
    const int Lm1 = L - 1;
    const uint64_t resmask = deposit_sites(bits, site_mask);
    const int w = !active ? (int)stride - 1 : (BOTH ? dir * N + s : s);   /* this lane's column of the rank lists (last: spare) */
    /* positive residue masses make every list ascending (what the neighbour-probe pairing needs) */
    const bool presorted = !__any(lane < L && !(m0 > 0.f && m1 > 0.f));
    /* every residue heavier than two tolerances (and a margin for the rounding of the running sums):
     * neighbouring ions of a list are more than two tolerances apart, so an ion has at most one partner
     * in the other list */
    const float wide_min = 2.f * cfg->mz_error + 0.02f;
    const bool wide = !__any(lane < L && !(m0 > wide_min && m1 > wide_min));
    wave_lds_sync();
    STAMP_T(b, 40, false);
    CumCounts cum = walk_record(f.resd, f.cum_lut, tab, L, zmax, resmask, dir, dir ? Ab : Af, dir ? Bb : Bf, Bf == 0. && Bb == 0.,
                                f.rkl, (int)stride, w);
    if (two_pass) {                                         /* ... and the same lane's backward walk (column N + s) */
        const CumCounts back = walk_record(f.resd, f.cum_lut, tab, L, zmax, resmask, 1, Ab, Bb, Bf == 0. && Bb == 0., f.rkl, (int)stride,
                                           active ? N + s : (int)stride - 1);
        cum.a += back.a;
        cum.b += back.b;
        cum.c += back.c;
    } else if (BOTH) {                                      /* a signature's two directions: the backward walker's counts to the forward one */
        uint32_t *r3 = f.rec + (size_t)s * 3;
        if (active && lane >= 32) {
            r3[0] = cum.a;
            r3[1] = cum.b;
            r3[2] = cum.c;
        }
        wave_lds_sync();
        if (active && lane < 32) {
            cum.a += r3[0];
            cum.b += r3[1];
            cum.c += r3[2];
        }
    }
    wave_lds_sync();
    STAMP_T(b, 41, false);

    const uint32_t nfrag = (uint32_t)ndir * (uint32_t)Lm1 * (uint32_t)zmax;     /* <= 255 (host) */
    int fail = 0;
    float ws = 0.f;
    if (active && (!BOTH || two_pass || lane < 32)) {
        /* scores from the cumulative counts (Ascore.cpp:123-139) */
        ws = -1.f;
        if (nfrag <= b.lut_n_max) {
            double sum = 0.;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                const float sc = lut_score(b, (uint32_t)d, cum.at(d), nfrag);
                const float prod = cfg->weights[d] * sc;                  /* float product ...   */
                sum = sum + (double)prod;                                 /* ... double sum      */
            }
            ws = (float)sum;
        } else {
            fail = 1;
        }
        uint32_t *r3 = f.rec + (size_t)s * 3;              /* (the counts as they are: a byte each) */
        r3[0] = cum.a;
        r3[1] = cum.b;
        r3[2] = cum.c;
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
    wave_lds_sync();                                        /* the count table and the peaks are free from here on */
    STAMP_T(b, 42, false);

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
        if ((uint32_t)lane < fused_site_cap(push_cap)) {
            f.site_max[lane] = 0;
            f.site_tie[lane] = 0;
            f.site_alt[lane] = 0ull;
        }
        if (__popcll(at_max) != 1 || (b.debug & 1024)) {
            const SortLds srt = sort_carve(f.sort_raw, N);
            if (sig_lane) {
                srt.key[lane] = my_ws;
                srt.idx[lane] = (uint16_t)lane;
            }
            wave_lds_sync();
            if (!(b.debug & 8) && sort_introsort_loop<true, true>(srt, N, true)) declined = true;
            const uint64_t m2 = __ballot(sig_lane && __float_as_uint(srt.key[lane]) == kmax);
            best_i = srt.idx[__builtin_ctzll(m2)];
        }
        best_ws = __uint_as_float(kmax);
        best_bits = __shfl(my_bits, (int)best_i, 64);
        wave_lds_sync();
    }
    STAMP_T(b, 43, false);
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
                atomicOr(&f.site_alt[a], 1ull << f.site_pos[__builtin_ctzll(came)]);
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
    STAMP_T(b, 44, false);
    float my_asc = __builtin_huge_valf();                   /* lane a keeps site a */
    const float err = cfg->mz_error;
    const FastDiv divL = fastdiv_make((uint32_t)(Lm1 > 0 ? Lm1 : 1));
    int p2 = 1;
    while (p2 <= Lm1) p2 <<= 1;                              /* the partner search covers indices 0 .. Lm1 */
    /* ---- Ascores, FUSED_ROUND competitors at a time ---- */
    for (uint32_t e0 = 0; !declined && e0 < np; e0 += FUSED_ROUND) {
        const int nc = (int)(np - e0) < FUSED_ROUND ? (int)(np - e0) : FUSED_ROUND;
        const int S = 1 + nc;
        /* fragment m/z of the winner (first round only) and of the round's competitors: one lane per
         * (signature, direction), the walk of walk_record without its lookups */
        if (lane < S * ndir && (e0 == 0 || lane >= ndir)) {
            const int sg = lane / ndir, d = lane - sg * ndir;
            const uint64_t sb = sg == 0 ? best_bits : f.pushed[e0 + sg - 1].bits;
            const int dd = BOTH ? d : (cfg->n_fwd > 0 ? 0 : 1);
            if (ZM) walk_mz_only(f.resd, L, zmax, deposit_sites(sb, site_mask), dd, dd ? Ab : Af, dd ? Bb : Bf,
                                 f.selm + (size_t)lane * ent_cap);
            else walk_sums_only(f.resd, L, deposit_sites(sb, site_mask), dd, f.selm + (size_t)lane * ent_cap);
        }
        if (!ZM) {
            /* charge 1: the few lanes above leave the running sums; every lane turns a share of them into m/z
             * (ModifiedPeptide.cpp:570-591), instead of the few doing the float64 arithmetic step by step */
            wave_lds_sync();
            const int l0 = e0 == 0 ? 0 : ndir, nl = S * ndir - l0;
            for (int i = lane; i < nl * Lm1; i += 64) {
                const int lq = (int)fastdiv((uint32_t)i, divL), st = i - lq * Lm1;
                const int dd = BOTH ? ((l0 + lq) & 1) : (cfg->n_fwd > 0 ? 0 : 1);
                float *at = f.selm + (size_t)(l0 + lq) * ent_cap + st;
                const double m = ((double)*at + (dd ? Ab : Af)) - (dd ? Bb : Bf);
                *at = (float)(m + 1.007825);
            }
        }
        /* depth scores of the winner and the competitors, read off the score table */
        for (int i = (e0 == 0 ? 0 : 10) + lane; i < S * 10; i += 64) {
            const int sg = i / 10, d = i - sg * 10;
            const uint32_t who = sg == 0 ? best_i : f.pushed[e0 + sg - 1].idx;
            const uint32_t cum = (f.rec[(size_t)who * 3 + (d >> 2)] >> ((d & 3) * 8)) & 0xffu;
            f.sc[i] = b.lut[lut_row(nfrag) + (uint32_t)d * (nfrag + 1) + cum];
        }
        wave_lds_sync();
        if (lane < nc) {
            const PushedEntry pe = f.pushed[e0 + lane];
            const uint64_t gone = best_bits & ~pe.bits, came = pe.bits & ~best_bits;
            const int a = __popcll(best_bits & (gone - 1));
            atomicOr(&f.site_alt[a], 1ull << f.site_pos[__builtin_ctzll(came)]);
            f.c_site[lane] = (uint32_t)a;
        }
        {
            /* depth of the largest score gap (Ascore.cpp:164-172: the first depth whose gap is the largest, 0 when no gap
             * is positive), one lane per (competitor, depth): a positive gap's bit pattern orders like its value, the
             * largest goes through an LDS maximum and a ballot names the first depth that has it */
            const int c = lane >> 4, d = lane & 15;
            const bool on = c < nc && d < PYA_NTOP;
            uint32_t key = 0;
            if (on) {
                const float diff = f.sc[d] - f.sc[(c + 1) * 10 + d];
                key = diff > 0.f ? __float_as_uint(diff) : 0u;
            }
            if (lane < nc) f.c_depth[lane] = 0;
            wave_lds_sync();
            if (on && key) atomicMax((uint32_t *)&f.c_depth[c], key);
            wave_lds_sync();
            const uint64_t at_max = __ballot(on && key && key == (uint32_t)f.c_depth[c]);
            wave_lds_sync();
            if (lane < nc) {
                const uint32_t m16 = (uint32_t)(at_max >> (16 * lane)) & 0xffffu;
                f.c_depth[lane] = m16 ? __builtin_ctz(m16) : 0;
            }
        }
        if (lane < nc * ndir * 2) {
            f.c_tr[lane] = 0;
            f.c_cnt[lane] = 0;
        }
        if (lane == 0) *f.bad_tasks = 0u;
        wave_lds_sync();
        STAMP_T(b, 45, false);
        /* ---- site-determining ions (cpp/ModifiedPeptide.cpp:259-320): an ion survives when the other
         * signature's list has no ion within mz_error of it.  The lists are ascending and
         * position-indexed, so the candidates sit at the ion's own index and its neighbours (a
         * binary search otherwise); with at most one partner per ion the reference's greedy walk
         * cancels exactly the partnered pairs (localize_core.hip.h); a task with a doubly partnered
         * ion is replayed below. ---- */
        const int ents = Lm1 * zmax;                        /* ions of one list: one ascending run per charge */
        const int items_all = nc * ndir * 2 * ents;
        const FastDiv divE = fastdiv_make((uint32_t)(ents > 0 ? ents : 1));
        /* Charge 1 and `wide` (above): the winner and a competitor differ in the modification state of some
         * residues; the fragments that contain none of them, or all of them, have the same residues in
         * both signatures -- their ions differ by rounding at most, pair with each other and (wide) with
         * nothing else, so they are not site-determining.  Only the steps in between are examined: from the
         * step that takes in the first differing residue to the one before the last (in travel order). */
        const bool spans = !ZM && wide && !(b.debug & 2048);
        /* Several fragment charges: a list is one ascending run per charge, and an ion's partners can sit in
         * any run of the other list.  Instead of probing every run and replaying a task serially when an ion
         * has two partners, both lists are merged by rank (below) and the reference's greedy walk runs per
         * cluster of ions chained by gaps below the tolerance, one lane per cluster (localize_core.hip.h,
         * same argument): one binary search per ion, no replay. */
        const bool zm_clusters = ZM && !(b.debug & 2048);
        int items = items_all;
        if (spans) {
            const int ntask = nc * ndir;
            if (lane < ntask) {
                const int c = lane / ndir, d = lane - c * ndir;
                const int dd = BOTH ? d : (cfg->n_fwd > 0 ? 0 : 1);
                const uint64_t diff = best_bits ^ f.pushed[e0 + c].bits;            /* site indices */
                const int r_lo = (int)f.site_pos[__builtin_ctzll(diff)];
                const int r_hi = (int)f.site_pos[63 - __builtin_clzll(diff)];
                const int lo = dd ? L - 1 - r_hi : r_lo, hi = dd ? L - 1 - r_lo : r_hi;
                f.t_lo[lane] = (uint32_t)lo;
                f.t_off[lane] = (uint32_t)(2 * (hi - lo));       /* both sides */
            }
            wave_lds_sync();
            uint32_t acc = 0;
            for (int t = 0; t < ntask; t++) {                 /* (at most FUSED_ROUND x 2 tasks: all lanes, same values) */
                const uint32_t n = f.t_off[t];
                wave_lds_sync();
                if (lane == 0) f.t_off[t] = acc;
                acc += n;
            }
            if (lane == 0) f.t_off[ntask] = acc;
            items = (int)acc;
            wave_lds_sync();
        }
        if (!(b.debug & 1) && !zm_clusters)
        for (int base = 0; base < items; base += 64) {
            const int e = base + lane;
            if (e < items) {
                int task, side, en;
                if (spans) {
                    const int ntask = nc * ndir;
                    task = 0;
                    for (int t = 1; t < ntask; t++) task += (uint32_t)e >= f.t_off[t] ? 1 : 0;
                    const int rem = e - (int)f.t_off[task];
                    const int len = (int)(f.t_off[task + 1] - f.t_off[task]) >> 1;
                    side = rem >= len ? 1 : 0;
                    en = (int)f.t_lo[task] + rem - side * len;
                } else {
                    const uint32_t ts = fastdiv((uint32_t)e, divE);
                    en = e - (int)ts * ents;
                    side = (int)ts & 1;
                    task = (int)ts >> 1;
                }
                const int z0 = zmax == 1 ? 0 : (int)fastdiv((uint32_t)en, divL), i = en - z0 * Lm1;
                const int d = BOTH ? task & 1 : 0;
                const int c = BOTH ? task >> 1 : task;
                const float *la = f.selm + (size_t)d * ent_cap;                            /* winner     */
                const float *lb = f.selm + (size_t)((1 + c) * ndir + d) * ent_cap;         /* competitor */
                const float *mine = side ? lb : la, *others = side ? la : lb;
                const float me = mine[en];
                int total = 0;                              /* ions of the other list within mz_error of this one */
                {
                    /* the run of the same charge: position-indexed like mine */
                    const float *other = others + z0 * Lm1;
                    float df[4];
                    bool ok[4], sk[4];
#pragma unroll
                    for (int uu = 0; uu < 4; uu++) {
                        const int q = i - 1 + uu;
                        ok[uu] = q >= 0 && q < Lm1;
                        const float o = ok[uu] ? other[q] : (q < 0 ? -__builtin_huge_valf() : __builtin_huge_valf());
                        /* (winner's ion) - (competitor's ion) is me - o on one side and o - me = -(me - o), exactly, on the
                         * other, where the walk skips the other list's ion when that difference is <= -err: either way the
                         * test is (me - o) >= err, and a partner is |me - o| < err */
                        df[uu] = me - o;
                        sk[uu] = df[uu] >= err;
                    }
                    const int w1 = (ok[1] && __builtin_fabsf(df[1]) < err) ? 1 : 0;
                    const int w2 = (ok[2] && __builtin_fabsf(df[2]) < err) ? 1 : 0;
                    const int w3 = (ok[3] && __builtin_fabsf(df[3]) < err) ? 1 : 0;
                    int cnt = -1;
                    if (sk[0] && !sk[1]) cnt = w1 + w2;      /* first candidate = index i     */
                    else if (sk[1] && !sk[2]) cnt = w2 + w3; /* first candidate = index i + 1 */
                    if (cnt < 0) cnt = partners_in_run(other, Lm1, p2, me, side, err);
                    total = cnt;
                }
                for (int zz = 0; zz < zmax; zz++)           /* the runs of the other charges: anywhere */
                    if (zz != z0) total += partners_in_run(others + zz * Lm1, Lm1, p2, me, side, err);
                if (total > 1 || (b.debug & 2048)) {
                    atomicOr(f.bad_tasks, 1u << task);      /* two partners: the serial walk decides */
                } else if (total == 0) {
                    const uint32_t who = side ? f.pushed[e0 + c].idx : best_i;
                    atomicAdd(&f.c_tr[task * 2 + side], 1u);
                    if ((int)f.rkl[(size_t)(i * zmax + z0) * stride + (d * N + (int)who)] <= f.c_depth[c])
                        atomicAdd(&f.c_cnt[task * 2 + side], 1u);
                }
            }
        }
        wave_lds_sync();
        const uint32_t bad = zm_clusters ? (1u << (nc * ndir)) - 1u : *f.bad_tasks;
        if (!ZM && bad) declined = true;                     /* (no room for a replay here: the general kernel takes it) */
        if (ZM && bad && !(b.debug & 1)) {
            /* ---- a task with a doubly partnered ion is replayed with the reference's serial walk over its
             * two sorted lists (ModifiedPeptide.cpp:291-316).  Sorting = merging the per-charge runs:
             * an ion's place is its index in its own run plus, for every other run, the number of ions
             * that sort before it (equal values: the lower charge first). ---- */
            for (int base = 0; base < items_all; base += 64) {
                const int e = base + lane;
                if (e < items_all) {
                    const uint32_t ts = fastdiv((uint32_t)e, divE);
                    const int en = e - (int)ts * ents;
                    const int z0 = zmax == 1 ? 0 : (int)fastdiv((uint32_t)en, divL), i = en - z0 * Lm1;
                    const int side = (int)ts & 1;
                    const int d = BOTH ? ((int)ts >> 1) & 1 : 0;
                    const int c = BOTH ? (int)ts >> 2 : (int)ts >> 1;
                    const int task = c * ndir + d;
                    if ((bad >> task) & 1u) {
                        const float *mine = f.selm + (size_t)((side ? (1 + c) * ndir : 0) + d) * ent_cap;
                        const float me = mine[en];
                        int pos = i;
                        for (int zz = 0; zz < zmax; zz++) {
                            if (zz == z0) continue;
                            const float *run = mine + zz * Lm1;
                            int j = 0;                          /* ions of run zz that sort before this one */
                            for (int step = p2 >> 1; step > 0; step >>= 1) {
                                const int probe = j + step;
                                if (probe - 1 < Lm1 && (zz < z0 ? run[probe - 1] <= me : run[probe - 1] < me)) j = probe;
                            }
                            pos += j;
                        }
                        const uint32_t who = side ? f.pushed[e0 + c].idx : best_i;
                        const size_t at = (size_t)(task * 2 + side) * ent_cap + pos;
                        f.srt_v[at] = me;
                        f.srt_h[at] = (int)f.rkl[(size_t)(i * zmax + z0) * stride + (d * N + (int)who)] <= f.c_depth[c] ? 1 : 0;
                    }
                }
            }
            wave_lds_sync();
            if (zm_clusters) {
                int p2e = 1;
                while (p2e <= ents) p2e <<= 1;
                for (int base = 0; base < items_all; base += 64) {
                    const int e = base + lane;
                    if (e >= items_all) continue;
                    const uint32_t ts = fastdiv((uint32_t)e, divE);     /* task * 2 + side, as the lists are laid out */
                    const int i = e - (int)ts * ents;
                    const int side = (int)ts & 1;
                    const int d = BOTH ? ((int)ts >> 1) & 1 : 0;
                    const int c = BOTH ? (int)ts >> 2 : (int)ts >> 1;
                    const int task = c * ndir + d;
                    const float *va = f.srt_v + (size_t)(task * 2) * ent_cap, *vb = va + ent_cap;
                    const uint8_t *ha = f.srt_h + (size_t)(task * 2) * ent_cap, *hb = ha + ent_cap;
                    const float *mine = side ? vb : va, *other = side ? va : vb;
                    const float me = mine[i];
                    int j = 0;                              /* ions of the other list before this one (A first on ties) */
                    for (int step = p2e >> 1; step > 0; step >>= 1) {
                        const int probe = j + step;
                        if (probe - 1 < ents && (side ? other[probe - 1] <= me : other[probe - 1] < me)) j = probe;
                    }
                    const float before_own = i > 0 ? mine[i - 1] : -__builtin_huge_valf();
                    const float before_other = j > 0 ? other[j - 1] : -__builtin_huge_valf();
                    if (!(me - before_own >= err && me - before_other >= err)) continue;
                    int ia = side ? j : i, ib = side ? i : j;
                    uint32_t tr0 = 0, tr1 = 0, n0 = 0, n1 = 0;
                    float reached = -__builtin_huge_valf();
                    for (bool first = true;; first = false) {
                        const float xa = ia < ents ? va[ia] : __builtin_huge_valf();
                        const float xb = ib < ents ? vb[ib] : __builtin_huge_valf();
                        const float next = xa <= xb ? xa : xb;
                        if (next == __builtin_huge_valf()) break;
                        if (!first && next - reached >= err) break;
                        if (__builtin_fabsf(xa - xb) < err) {       /* ModifiedPeptide.cpp:291-316 */
                            ia++;
                            ib++;
                            reached = __builtin_fmaxf(reached, xa > xb ? xa : xb);
                        } else if (xa < xb) {
                            tr0++;
                            n0 += ha[ia++];
                            reached = __builtin_fmaxf(reached, xa);
                        } else {
                            tr1++;
                            n1 += hb[ib++];
                            reached = __builtin_fmaxf(reached, xb);
                        }
                    }
                    if (tr0) atomicAdd(&f.c_tr[task * 2], tr0);
                    if (tr1) atomicAdd(&f.c_tr[task * 2 + 1], tr1);
                    if (n0) atomicAdd(&f.c_cnt[task * 2], n0);
                    if (n1) atomicAdd(&f.c_cnt[task * 2 + 1], n1);
                }
            } else
            if (lane < nc * ndir && ((bad >> lane) & 1u)) {
                const float *va = f.srt_v + (size_t)(lane * 2) * ent_cap, *vb = va + ent_cap;
                const uint8_t *ha = f.srt_h + (size_t)(lane * 2) * ent_cap, *hb = ha + ent_cap;
                uint32_t tr0 = 0, tr1 = 0, n0 = 0, n1 = 0;
                int ia = 0, ib = 0;
                while (ia < ents || ib < ents) {
                    if (ib == ents) {
                        tr0++;
                        n0 += ha[ia++];
                    } else if (ia == ents) {
                        tr1++;
                        n1 += hb[ib++];
                    } else {
                        const float xa = va[ia], xb = vb[ib];
                        if (__builtin_fabsf(xa - xb) < err) {
                            ia++;
                            ib++;
                        } else if (xa < xb) {
                            tr0++;
                            n0 += ha[ia++];
                        } else {
                            tr1++;
                            n1 += hb[ib++];
                        }
                    }
                }
                f.c_tr[lane * 2] = tr0;
                f.c_tr[lane * 2 + 1] = tr1;
                f.c_cnt[lane * 2] = n0;
                f.c_cnt[lane * 2 + 1] = n1;
            }
            wave_lds_sync();
        }
        STAMP_T(b, 46, false);
        if (!declined) {
            /* ---- Ascores (cpp/Ascore.cpp:200-209, :239-251, :305-313) ---- */
            float asc_l = 0.f;
            if (lane < nc) {
                uint32_t tr0 = 0, tr1 = 0, n0 = 0, n1 = 0;
                for (int d = 0; d < ndir; d++) {            /* a competitor's tasks: one per direction */
                    const int t = lane * ndir + d;
                    tr0 += f.c_tr[t * 2];
                    tr1 += f.c_tr[t * 2 + 1];
                    n0 += f.c_cnt[t * 2];
                    n1 += f.c_cnt[t * 2 + 1];
                }
                const uint32_t depth = (uint32_t)f.c_depth[lane];
                if (tr0 > b.lut_n_max || tr1 > b.lut_n_max) {
                    fail = 1;
                } else {
                    const float sc0 = b.lut[lut_row(tr0) + depth * (tr0 + 1) + n0];
                    const float sc1 = b.lut[lut_row(tr1) + depth * (tr1 + 1) + n1];
                    asc_l = sc0 - sc1;
                }
            }
            for (int c = 0; c < nc; c++) {
                const float asc = __shfl(asc_l, c, 64);
                if (lane == (int)f.c_site[c]) my_asc = asc < my_asc ? asc : my_asc;
            }
        }
        wave_lds_sync();
    }
    if (declined) {
        /* ---- hand-over: leave what score_signatures would have left ---- */
        if (sig_lane) {
            b.ws[s0 + lane] = my_ws;
            if (b.rec) {
                const uint32_t *r3 = f.rec + (size_t)lane * 3;
                uint32_t *dst = b.rec + (s0 + lane) * PYA_REC_WORDS;
                uint32_t cum[PYA_NTOP];
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) cum[d] = (r3[d >> 2] >> ((d & 3) * 8)) & 0xffu;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) dst[d >> 1] = cum[d] | (cum[d + 1] << 16);
                dst[5] = nfrag;
            }
        }
        ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)f.grid)[lane];
        if (lane == 0) b.ws_top[(size_t)psm * 4 + 1] = 0u;  /* no summary of the scores: localize scans them */
        return true;
    }
    STAMP_T(b, 47, false);
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
