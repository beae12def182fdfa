/* localize_hash.hip.h -- site-determining ions of the general route (several fragment charges, neutral
 * losses, any number of ion types) without generating, sorting and walking whole fragment lists.
 *
 * ModifiedPeptide::getSiteDeterminingIons (cpp/ModifiedPeptide.cpp:259-320) sorts every fragment of the two
 * site assignments (all charges, all loss variants: ~230 each on cfg4) and cancels them with a greedy two-pointer
 * walk.  The winner (list A) and a single-move competitor (list B) differ in the residues between the two sites
 * only.  A prefix outside that span has the same loss variants in both and float32 running sums that differ
 * by rounding, so each of its A ions has a twin in B a few ulps away -- and a run of the merged list made of
 * twin pairs only cancels completely whatever the order inside it: the walk enters it with both cursors past
 * everything earlier (the restart property localize_core.hip.h relies on: an ion mz_error or more above every
 * earlier ion), the sorted A and the sorted B members are then pairwise within a few ulps, and every
 * comparison cancels.  So only *runs that contain an ion of a differing prefix* can leave anything:
 *   - prefixes are "in span" for (competitor, direction) when their loss variants differ or their running
 *     sums differ by more than tau = mz_error / 8 (so the span needs no assumption about where it is);
 *   - every in-span ion (both sides; ~60 of 230 on cfg4) asks a hash grid over the winner's ions and the
 *     competitor's in-span ions whether anything lies within mz_error + tau + rounding of it.  Nothing
 *     there (99.4 % on cfg4): the ion is a run of its own and is site-determining;
 *   - otherwise the run around it is gathered exactly (twins computed, not assumed), delimited with the
 *     walk's own arithmetic, and walked by the wavefront; the lowest in-span ion of a run does the
 *     accounting for all of it.
 * Nothing is sorted.  What does not fit the fixed-size tables is declined (the caller hands the PSM to the
 * list-based instantiation).  PYA_DEBUG bits: 8192 decline everything, 16384 every in-span ion takes the
 * exact route. */
#ifndef PYA_LOCALIZE_HASH_H
#define PYA_LOCALIZE_HASH_H
/* (included by localize_core.hip.h, after loc_site_ions) */

#define LH_SLOW_CAP 64
#define LH_LISTS (2 * PYA_LOC_SB_MAX)     /* 1 + 2 per competitor, rounded up to even */

DEV uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t w = (uint64_t)__shfl_xor((long long)v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}

/* work area (where the list-based route keeps its fragment pool): sizes are launch parameters */
struct HashLds {
    float *val;            /* [vc] ions of the current (type, competitor group): the winner's, then in-span B */
    uint32_t *tab;         /* [hs / 2] open-addressing table of 16-bit slots (entry + 1), two per word        */
    uint16_t *pairs;       /* [pp] (prefix | variant << 8) / winner pair indices, lists back to back          */
    uint32_t *slow;        /* [LH_SLOW_CAP] items for the exact route: entry | competitor << 16 | side << 24  */
    float *cand_val;       /* [LH_CAND_CAP] */
    uint32_t *cand_tag;    /* [LH_CAND_CAP] entry | in span << 16 | side << 17 */
    uint64_t *ispan;       /* [sb] in-span prefixes per competitor, current direction                         */
    uint16_t *off, *cnt;   /* [LH_LISTS] pair lists of the current direction: W = 0, A(c) = 1 + 2 (c - 1), B(c) = 2 + 2 (c - 1) */
    uint32_t vc, hs, pp;
};

#define LH_CAND_CAP 32
static inline __host__ __device__ size_t loc_hash_words(uint32_t vc, uint32_t hs, uint32_t pp, uint32_t sb) {
    return (size_t)vc + hs / 2 + (pp + 1) / 2 + LH_SLOW_CAP + 2 * LH_CAND_CAP + 2 * sb + LH_LISTS + 2;
}

DEV HashLds hash_carve(float *base, uint32_t vc, uint32_t hs, uint32_t pp, uint32_t sb) {
    HashLds h;
    h.ispan = (uint64_t *)base;                                  /* (the pool starts 8-byte aligned) */
    h.val = (float *)(h.ispan + sb);
    h.tab = (uint32_t *)(h.val + vc);
    h.slow = h.tab + hs / 2;
    h.cand_val = (float *)(h.slow + LH_SLOW_CAP);
    h.cand_tag = (uint32_t *)(h.cand_val + LH_CAND_CAP);
    h.off = (uint16_t *)(h.cand_tag + LH_CAND_CAP);
    h.cnt = h.off + LH_LISTS;
    h.pairs = h.cnt + LH_LISTS;
    h.vc = vc;
    h.hs = hs;
    h.pp = pp;
    return h;
}

DEV float lh_ion(float run, float loss, bool nn, double A, double B, int z) {
    const float x = nn ? run - loss : run;
    const double m = ((double)x + A) - B;
    return charge_mz(m, z);
}

DEV int lh_cell(float x, float inv_cw) { return (int)__builtin_floorf(x * inv_cw); }

DEV void lh_insert(const HashLds &h, int cell, int id, int hshift) {
    const uint32_t hmask = h.hs - 1u;
    uint32_t s = ((uint32_t)cell * 0x9E3779B1u) >> hshift;
    /* (a loop per lane: the same as one loop for the wavefront with selects -- what the probes below gain from -- measured
     * 0.19 ms slower on cfg4: a lane that lost a race for a slot holds the others up) */
    for (;;) {
        uint32_t *wp = h.tab + (s >> 1);
        const uint32_t sh = (s & 1u) * 16u;
        /* (an atomic load, not a volatile one: a volatile access through this pointer is compiled as a FLAT load with
         * system scope, which also waits for every global load in flight) */
        const uint32_t old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if ((old >> sh) & 0xffffu) {
            s = (s + 1u) & hmask;
            continue;
        }
        if (atomicCAS(wp, old, old | ((uint32_t)(id + 1) << sh)) == old) break;
    }
}

/* is an entry of this task (the winner's ions, the competitor's in-span ions [b_lo, b_hi)) other than `self` within `reach`
 * of x?  Every entry whose cell is k sits in the probe run that starts at k's hash; the (one or two) cells an ion's window
 * touches are visited one after the other by the same loop.  ONE loop for the wavefront, every lane's state advanced by
 * selects: as a loop with two exits per lane the exec-mask bookkeeping cost as many scalar instructions as the body has
 * vector ones (cfg4 localize 6.24 -> 6.01 ms). */
DEV bool lh_probe2(const HashLds &h, int k0, int k1, int hshift, int self, float x, float reach, int nW, int b_lo, int b_hi) {
    const uint32_t hmask = h.hs - 1u;
    const uint16_t *t16 = (const uint16_t *)h.tab;
    uint32_t s = ((uint32_t)k0 * 0x9E3779B1u) >> hshift;
    const uint32_t s1 = ((uint32_t)k1 * 0x9E3779B1u) >> hshift;
    uint32_t left = k1 == k0 ? 1u : 2u;                      /* probe runs still to finish */
    uint32_t hit = 0u;
    while (__any(left != 0u)) {
        const uint32_t half = t16[s];
        const bool empty = half == 0u;
        const int j = (int)(empty ? 1u : half) - 1;
        const float dist = __builtin_fabsf(h.val[j] - x);
        const bool rel = left != 0u && !empty && (j != self) && (j < nW || (j >= b_lo && j < b_hi));
        hit |= (rel && dist < reach) ? 1u : 0u;
        s = empty ? s1 : ((s + 1u) & hmask);
        left = (empty && left != 0u) ? left - 1u : left;
    }
    return hit != 0u;
}

/* the same, looking a little further: near = the one such entry within `rf`, multi = there is more than one.  (An entry can
 * sit in both cells' probe runs: entries are told apart by their number, not counted.) */
DEV void lh_probe_near(const HashLds &h, int k0, int k1, int hshift, int self, float x, float rf, int nW, int b_lo, int b_hi,
                       int &near, bool &multi) {
    const uint32_t hmask = h.hs - 1u;
    const uint16_t *t16 = (const uint16_t *)h.tab;
    uint32_t s = ((uint32_t)k0 * 0x9E3779B1u) >> hshift;
    const uint32_t s1 = ((uint32_t)k1 * 0x9E3779B1u) >> hshift;
    uint32_t left = k1 == k0 ? 1u : 2u;
    uint32_t mul = 0u;
    while (__any(left != 0u)) {
        const uint32_t half = t16[s];
        const bool empty = half == 0u;
        const int j = (int)(empty ? 1u : half) - 1;
        const float dist = __builtin_fabsf(h.val[j] - x);
        const bool in_f = left != 0u && !empty && (j != self) && (j < nW || (j >= b_lo && j < b_hi)) && dist < rf;
        mul |= (in_f && near >= 0 && near != j) ? 1u : 0u;
        near = in_f ? j : near;
        s = empty ? s1 : ((s + 1u) & hmask);
        left = (empty && left != 0u) ? left - 1u : left;
    }
    multi |= mul != 0u;
}

DEV uint32_t lh_ord(float v) {                          /* float -> unsigned with the same order */
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* Surviving ions wait in a staging buffer (128 values + 128 one-byte tags) and are looked up 64 at a time, one
 * per lane, in the workspace table.  (Staging the table in LDS cost more occupancy than the lookups cost time;
 * two lookups per lane with overlapped round trips, and no lookups at all -- PYA_DEBUG=2 -- leave the kernel's
 * time unchanged: it is bound by issue at the occupancy its LDS allows, not by these loads.) */
#ifndef LH_CELL_QR
#define LH_CELL_QR 16.f
#endif
#define LH_STAGE 128
/* The buffer is a ring: `staged` = (entries waiting) | (slot of the oldest << 16).  A flush looks the oldest 64 (or fewer) up
 * and moves the head; nothing is copied to make room (r04 moved the remainder to the front after every flush). */
DEV void lh_stage_flush(const LocCtx &c, int &staged) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    float *sv = w.stage_val;
    uint8_t *st = (uint8_t *)(sv + LH_STAGE);
    wave_lds_sync();
    const int n = staged & 0xffff, head = staged >> 16;
    const int take = n < 64 ? n : 64;
    if (lane < take) {
        const int slot = (head + lane) & (LH_STAGE - 1);
        const uint32_t tg = st[slot];
        const int r = (c.b->debug & 2u) ? PYA_NO_MATCH : match_rank(c.tab, sv[slot]);     /* (2: ablation, no lookups) */
        /* ONE atomic per ion: trials in the low half of c_tr[tag], matches in its high half (loc_site_ions_hash unpacks them at
         * its end; a competitor has < 2^16 site-determining ions: <= 8 ion types x 2 048 fragments).  64 lanes add to a handful
         * of words here, and LDS atomics on one address take their turns (r05 counters: SQ_LDS_ADDR_CONFLICT = 16 % of the
         * kernel's LDS cycles with two atomics per ion) */
        atomicAdd(&w.c_tr[tg], 1u | (r <= w.c_depth[tg >> 1] ? 0x10000u : 0u));
    }
    staged = (n - take) | (((head + take) & (LH_STAGE - 1)) << 16);
    wave_lds_sync();
}
DEV void lh_stage_push(const LocCtx &c, bool kept, float val, uint32_t tag, int &staged) {
    const LocLds &w = c.w;
    const uint64_t km = __ballot(kept);
    const int n = staged & 0xffff, head = staged >> 16;
    if (kept) {
        const int slot = (head + n + mask_rank(km)) & (LH_STAGE - 1);
        w.stage_val[slot] = val;
        ((uint8_t *)(w.stage_val + LH_STAGE))[slot] = (uint8_t)tag;
    }
    staged = (n + __popcll(km)) | (head << 16);
    if ((staged & 0xffff) >= 64) lh_stage_flush(c, staged);
}

/* what the exact route needs to know about the table in place */
struct LhPass {
    int d, zmax, c0, c1, nW, offW, hshift;
    double A, B;
    float err, margin, R, rf, qf, inv_cw;
    bool nn;
    FastDiv divZ;
};
#define LH_HIT_CAP (LH_SLOW_CAP + 2 * LH_CAND_CAP)      /* the list of ions with a neighbour: h.slow and, until the exact route runs, its candidate arrays behind it */

/* The exact route for the items of h.slow[0, nslow): the run of the merged list around each item is gathered
 * (the winner's ions, the twins of those outside the span computed with the competitor's running sums, the
 * competitor's in-span ions), delimited with the walk's own restart test and walked (ModifiedPeptide.cpp:291-316)
 * by the wavefront, one candidate per lane.  The lowest in-span ion of a run accounts for the whole run.
 * Returns true when the PSM has to be declined. */
DEV bool lh_exact(const LocCtx &c, const HashLds &h, const LhPass &q, uint32_t nslow, int &staged) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const float err = q.err, R = q.R;
    const int zmax = q.zmax, d = q.d, nW = q.nW;
    wave_lds_sync();
    for (uint32_t k = 0; k < nslow; k++) {
        const uint32_t it = h.slow[k];
        const int id0 = (int)(it & 0xffffu), cs = (int)((it >> 16) & 0xffu), side0 = (int)(it >> 24);
        const float x0 = h.val[id0];
        const uint64_t ins = h.ispan[cs - 1];
        int bl = nW;
        for (int cc = q.c0; cc < cs; cc++) bl += (int)h.cnt[2 + 2 * (cc - 1)] * zmax;
        const int bh = bl + (int)h.cnt[2 + 2 * (cs - 1)] * zmax;
        int ncand = 0;
        wave_lds_sync();
        for (int base = 0; base < nW; base += 64) {
            const int j = base + lane;
            bool a_in = false, t_in = false;
            float av = 0.f, tv = 0.f;
            uint32_t isp = 0;
            if (j < nW) {
                av = h.val[j];
                const float da = __builtin_fabsf(av - x0);
                if (da < R + q.margin) {
                    const int pair = (int)fastdiv((uint32_t)j, q.divZ), z = j - pair * zmax + 1;
                    const uint32_t pz = h.pairs[q.offW + pair];
                    const int p = (int)(pz & 255u), v = (int)(pz >> 8);
                    isp = (uint32_t)((ins >> p) & 1ull);
                    a_in = da < R;
                    if (!isp) {
                        tv = lh_ion(w.run[(size_t)(cs * 2 + d) * c.pos_cap + p], q.nn ? c.nl.uniq[v] : 0.f, q.nn, q.A, q.B, z);
                        t_in = __builtin_fabsf(tv - x0) < R;
                    }
                }
            }
            const uint64_t ma = __ballot(a_in);
            if (a_in) {
                const int slot = ncand + mask_rank(ma);
                if (slot < LH_CAND_CAP) {
                    h.cand_val[slot] = av;
                    h.cand_tag[slot] = (uint32_t)j | (isp << 16);
                }
            }
            ncand += __popcll(ma);
            const uint64_t mt = __ballot(t_in);
            if (t_in) {
                const int slot = ncand + mask_rank(mt);
                if (slot < LH_CAND_CAP) {
                    h.cand_val[slot] = tv;
                    h.cand_tag[slot] = (uint32_t)j | (1u << 17);
                }
            }
            ncand += __popcll(mt);
        }
        for (int base = bl; base < bh; base += 64) {
            const int j = base + lane;
            float bv = 0.f;
            bool b_in = false;
            if (j < bh) {
                bv = h.val[j];
                b_in = __builtin_fabsf(bv - x0) < R;
            }
            const uint64_t mb = __ballot(b_in);
            if (b_in) {
                const int slot = ncand + mask_rank(mb);
                if (slot < LH_CAND_CAP) {
                    h.cand_val[slot] = bv;
                    h.cand_tag[slot] = (uint32_t)j | (1u << 16) | (1u << 17);
                }
            }
            ncand += __popcll(mb);
        }
        if (ncand > LH_CAND_CAP) return true;
        wave_lds_sync();
        const bool has = lane < ncand;
        const float v = has ? h.cand_val[lane] : 0.f;
        const uint32_t tag = has ? h.cand_tag[lane] : 0u;
        const int side = (int)((tag >> 17) & 1u);
        const bool isp = ((tag >> 16) & 1u) != 0u;
        /* the run: everything chained to the item by gaps the walk does not restart at */
        float lo = x0, hi = x0;
        for (;;) {
            const float up = (has && v > hi && !(v - hi >= err)) ? v : hi;
            const float dn = (has && v < lo && !(lo - v >= err)) ? v : lo;
            const float nh = wave_max_f32(up), nl = wave_min_f32(dn);
            if (nh == hi && nl == lo) break;
            hi = nh;
            lo = nl;
        }
        /* (an ion outside the window is R or more from the item: it must not be able to chain to the run) */
        if (!(R - (hi - x0) >= 2.f * err) || !(R - (x0 - lo) >= 2.f * err)) return true;
        const bool member = has && v >= lo && v <= hi;
        const uint64_t key = (member && isp) ? (((uint64_t)lh_ord(v) << 32) | ((uint64_t)side << 16) | (uint64_t)(tag & 0xffffu)) : ~0ull;
        const uint64_t own = ((uint64_t)lh_ord(x0) << 32) | ((uint64_t)side0 << 16) | (uint64_t)id0;
        if (wave_min_u64(key) != own) continue;          /* a lower in-span ion of the run accounts for it */
        bool remA = member && side == 0, remB = member && side == 1, kept = false;
        const float inf = __builtin_huge_valf();
        for (;;) {
            const float xa = wave_min_f32(remA ? v : inf), yb = wave_min_f32(remB ? v : inf);
            if (xa == inf && yb == inf) break;
            bool takeA, takeB, keep = true;
            if (yb == inf) {
                takeA = true;
                takeB = false;
            } else if (xa == inf) {
                takeA = false;
                takeB = true;
            } else if (__builtin_fabsf(xa - yb) < err) {
                takeA = takeB = true;
                keep = false;
            } else if (xa < yb) {
                takeA = true;
                takeB = false;
            } else {
                takeA = false;
                takeB = true;
            }
            if (takeA) {
                const int la = __builtin_ctzll(__ballot(remA && v == xa));
                if (lane == la) {
                    remA = false;
                    kept = keep;
                }
            }
            if (takeB) {
                const int lb = __builtin_ctzll(__ballot(remB && v == yb));
                if (lane == lb) {
                    remB = false;
                    kept = keep;
                }
            }
        }
        lh_stage_push(c, kept, v, (uint32_t)(cs * 2 + side), staged);
    }
    wave_lds_sync();
    return false;
}

/* The ions the query loop found a neighbour for (h.slow[0, nslow), nslow <= LH_HIT_CAP), one per lane.  Nearly all of them
 * are one of two simple cases, decided here in closed form; what is left goes through the exact route.
 *
 * (a) x next to ONE ion y of the winner outside the span -- which has a twin y' in the competitor's list (|y - y'| <=
 *     margin).  When nothing else of the task lies within rf = 4 err of x, nothing else can pair with or chain to
 *     {x, y, y'}: others are >= rf, their twins >= rf - margin away, the three lie within reach + margin of x, and
 *     3 err - 3 margin - 2 slop >= err under the caller's guard.  The walk (ModifiedPeptide.cpp:291-316) over A = {x, y} /
 *     B = {y'} (x the winner's) or A = {y} / B = {x, y'} (x the competitor's): with s the pair's ion on x's side and o the
 *     other one, x <= s and |x - o| < err pairs x with o and leaves s; otherwise x is taken (it is below o, or above the
 *     pair, which pairs off).
 * (b) x next to ONE other ion y of the span (the winner's or the competitor's: no twins).  Both are items of the query loop,
 *     and the exact route accounts for a run as a whole, so both must take the same way: y has to be as alone as x is --
 *     nothing but x within rf of it (each lane checks both, so both reach the same verdict).  Then the walk pairs them off
 *     when they are on opposite sides and closer than err, and takes both otherwise.
 * 58 % / 40 % of cfg4's collisions; the exact route, serial per ion, was 2.05 of the kernel's 7.7 ms. */
DEV bool lh_resolve(const LocCtx &c, const HashLds &h, const LhPass &q, uint32_t nslow, int &staged) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const int zmax = q.zmax, nW = q.nW;
    wave_lds_sync();
    /* (the second half of the list lies where the exact route keeps its candidates: into registers first) */
    const uint32_t n_lo = nslow < (uint32_t)LH_SLOW_CAP ? nslow : (uint32_t)LH_SLOW_CAP;
    const uint32_t it_lo = (uint32_t)lane < n_lo ? h.slow[lane] : 0u;
    const uint32_t it_hi = (uint32_t)(LH_SLOW_CAP + lane) < nslow ? h.slow[LH_SLOW_CAP + lane] : 0u;
    const bool closed = !(c.b->debug & (0x20000000u | 16384u));
    for (int chunk = 0; chunk < 2; chunk++) {
        const uint32_t n = chunk ? nslow - n_lo : n_lo;
        if (n == 0) break;
        const uint32_t it = chunk ? it_hi : it_lo;
        const bool on = (uint32_t)lane < n;
        const int id = (int)(it & 0xffffu), cc = on ? (int)((it >> 16) & 0xffu) : q.c0, side = (int)(it >> 24);
        bool unresolved = on, kept = false;
        float x = 0.f;
        if (on && closed) {
            int b_lo = nW;
            for (int q2 = q.c0; q2 < q.c1; q2++) b_lo += q2 < cc ? (int)h.cnt[2 + 2 * (q2 - 1)] * zmax : 0;
            const int b_hi = b_lo + (int)h.cnt[2 + 2 * (cc - 1)] * zmax;
            x = h.val[id];
            int near = -1;
            bool multi = false;
            lh_probe_near(h, lh_cell(x - q.qf, q.inv_cw), lh_cell(x + q.qf, q.inv_cw), q.hshift, id, x, q.rf, nW, b_lo, b_hi, near, multi);
            if (near >= 0 && !multi) {
                /* one neighbour.  An ion of the winner outside the span has a twin; one in the span, or an ion of the competitor
                 * (all of those in the table are in the span), has not */
                int p = 0, v = 0, z = 1;
                bool y_in_span = near >= nW;
                if (!y_in_span) {
                    const int pair = (int)fastdiv((uint32_t)near, q.divZ);
                    const uint32_t pz = h.pairs[q.offW + pair];
                    z = near - pair * zmax + 1;
                    p = (int)(pz & 255u);
                    v = (int)(pz >> 8);
                    y_in_span = ((h.ispan[cc - 1] >> p) & 1ull) != 0ull;
                }
                const float y = h.val[near];
                if (y_in_span) {
                    int near2 = -1;
                    bool multi2 = false;
                    lh_probe_near(h, lh_cell(y - q.qf, q.inv_cw), lh_cell(y + q.qf, q.inv_cw), q.hshift, near, y, q.rf, nW, b_lo, b_hi,
                                  near2, multi2);
                    if (near2 == id && !multi2) {
                        const int y_side = near >= nW ? 1 : 0;
                        kept = !(y_side != side && __builtin_fabsf(x - y) < q.err);
                        unresolved = false;
                    }
                } else {
                    const float yt = lh_ion(w.run[(size_t)(cc * 2 + q.d) * c.pos_cap + p], q.nn ? c.nl.uniq[v] : 0.f, q.nn, q.A, q.B, z);
                    const float sv = side ? yt : y, ov = side ? y : yt;
                    if (x <= sv && __builtin_fabsf(x - ov) < q.err) x = sv;
                    kept = true;
                    unresolved = false;
                }
#ifdef PYA_STAMPS
                if (c.b->stamps) atomicAdd(&c.b->stamps[unresolved ? 44 : (y_in_span ? 42 : 41)], 1ull);   /* (diagnostic build) */
#endif
            }
#ifdef PYA_STAMPS
            if (c.b->stamps) atomicAdd(&c.b->stamps[40], 1ull);
            if (c.b->stamps && multi) atomicAdd(&c.b->stamps[43], 1ull);
#endif
        }
        wave_lds_sync();                                         /* (everybody has read its entry) */
        const uint64_t um = __ballot(unresolved);
        if (unresolved) h.slow[mask_rank(um)] = it;
        lh_stage_push(c, kept, x, (uint32_t)(cc * 2 + side), staged);
        const uint32_t nleft = (uint32_t)__popcll(um);
        if (nleft && lh_exact(c, h, q, nleft, staged)) return true;
    }
    return false;
}

/* Pair lists of one direction (built when the ion types change direction: one direction's lists at a time halve
 * the room they need): list 0 = the winner's (prefix | variant << 8) pairs; per competitor cc the in-span prefixes
 * (h.ispan[cc - 1]: loss variants differ, or running sums differ by more than tau), list 1 + 2 (cc - 1) = the
 * winner's pair indices on them (A), list 2 + 2 (cc - 1) = the competitor's pairs on them (B).  True = no room. */
DEV bool lh_build_lists(const LocCtx &c, const HashLds &h, int S, int d, float tau, bool nn) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const int Lm1 = c.L - 1;
    uint32_t npairs = 0;
    {
        uint32_t pw = 0;
        if (lane < Lm1) pw = nn ? (uint32_t)w.pmk[(size_t)d * c.pos_cap + lane] : 1u;
        int tot;
        int at = wave_excl_scan_i32(__popc(pw), &tot);
        if ((uint32_t)tot > h.pp) return true;
        while (pw) {
            const int v = __builtin_ctz(pw);
            pw &= pw - 1;
            h.pairs[at++] = (uint16_t)(lane | (v << 8));
        }
        if (lane == 0) {
            h.off[0] = 0;
            h.cnt[0] = (uint16_t)tot;
        }
        npairs = (uint32_t)tot;
    }
#ifndef LH_LISTS_PER_COMPETITOR
    if ((S - 1) * Lm1 <= 64) {
        /* every competitor's prefixes in ONE pass, a lane per (competitor, prefix) -- cfg4: 3 x 19 lanes where three passes
         * used 19 each: the two prefix sums run over the whole wavefront, a competitor's share of them is the difference of
         * the sums at the ends of its lanes (read with a wave-uniform lane index) */
        const int ci = Lm1 > 0 ? lane / Lm1 : 0, p = lane - ci * Lm1;
        const bool valid = ci < S - 1 && Lm1 > 0;
        bool in = false;
        uint32_t pw = 0, pc = 0;
        int cw = 0;
        if (valid) {
            const size_t iw = (size_t)d * c.pos_cap + p, ic = (size_t)((ci + 1) * 2 + d) * c.pos_cap + p;
            const float rw = w.run[iw], rc = w.run[ic];
            pw = nn ? (uint32_t)w.pmk[iw] : 1u;
            pc = nn ? (uint32_t)w.pmk[ic] : 1u;
            cw = nn ? (int)w.cpre[iw] : p;
            in = pw != pc || !(__builtin_fabsf(rc - rw) <= tau);
        }
        const uint64_t m_all = __ballot(in);
        const int nA = in ? __popc(pw) : 0, nB = in ? __popc(pc) : 0;
        const int sA = (int)wave_incl_scan_u32<false>((uint32_t)nA), sB = (int)wave_incl_scan_u32<false>((uint32_t)nB);
        int myA0 = 0, myB0 = 0, myOffA = 0, myOffB = 0, prevA = 0, prevB = 0;
        for (int q = 0; q < S - 1; q++) {
            const int e = (q + 1) * Lm1 - 1;
            const int endA = __builtin_amdgcn_readlane(sA, e), endB = __builtin_amdgcn_readlane(sB, e);
            const int totA = endA - prevA, totB = endB - prevB;
            if (ci == q) {
                myA0 = prevA;
                myB0 = prevB;
                myOffA = (int)npairs;
                myOffB = (int)npairs + totA;
            }
            if (lane == 0) {
                h.ispan[q] = (m_all >> (q * Lm1)) & ((1ull << Lm1) - 1ull);
                h.off[1 + 2 * q] = (uint16_t)npairs;
                h.cnt[1 + 2 * q] = (uint16_t)totA;
                h.off[2 + 2 * q] = (uint16_t)(npairs + (uint32_t)totA);
                h.cnt[2 + 2 * q] = (uint16_t)totB;
            }
            npairs += (uint32_t)(totA + totB);
            prevA = endA;
            prevB = endB;
        }
        if (npairs > h.pp) return true;
        if (in) {
            int atA = myOffA + (sA - nA - myA0), atB = myOffB + (sB - nB - myB0);
            for (int vi = 0; vi < nA; vi++) h.pairs[atA++] = (uint16_t)(cw + vi);
            while (pc) {
                const int v = __builtin_ctz(pc);
                pc &= pc - 1;
                h.pairs[atB++] = (uint16_t)(p | (v << 8));
            }
        }
        return false;
    }
#endif
    for (int cc = 1; cc < S; cc++) {
        bool in = false;
        uint32_t pw = 0, pc = 0;
        int cw = 0;
        if (lane < Lm1) {
            const size_t iw = (size_t)d * c.pos_cap + lane, ic = (size_t)(cc * 2 + d) * c.pos_cap + lane;
            const float rw = w.run[iw], rc = w.run[ic];
            pw = nn ? (uint32_t)w.pmk[iw] : 1u;
            pc = nn ? (uint32_t)w.pmk[ic] : 1u;
            cw = nn ? (int)w.cpre[iw] : lane;
            in = pw != pc || !(__builtin_fabsf(rc - rw) <= tau);
        }
        const uint64_t m = __ballot(in);
        int totA, totB;
        int atA = wave_excl_scan_i32(in ? __popc(pw) : 0, &totA);
        int atB = wave_excl_scan_i32(in ? __popc(pc) : 0, &totB);
        if (npairs + (uint32_t)(totA + totB) > h.pp) {
            return true;
        }
        atA += (int)npairs;
        atB += (int)npairs + totA;
        if (in) {
            for (int vi = 0, n = __popc(pw); vi < n; vi++) h.pairs[atA++] = (uint16_t)(cw + vi);
            while (pc) {
                const int v = __builtin_ctz(pc);
                pc &= pc - 1;
                h.pairs[atB++] = (uint16_t)(lane | (v << 8));
            }
        }
        if (lane == 0) {
            h.ispan[cc - 1] = m;
            h.off[1 + 2 * (cc - 1)] = (uint16_t)npairs;
            h.cnt[1 + 2 * (cc - 1)] = (uint16_t)totA;
            h.off[2 + 2 * (cc - 1)] = (uint16_t)(npairs + (uint32_t)totA);
            h.cnt[2 + 2 * (cc - 1)] = (uint16_t)totB;
        }
        npairs += (uint32_t)(totA + totB);
    }
    return false;
}

/* Site-determining ions of competitors 1..S-1 against the winner (entry 0): fills w.c_cnt / w.c_tr and the
 * depth in w.c_depth, as loc_site_ions does.  Returns true when the PSM is declined (nothing to undo: the
 * caller writes no result). */
DEV bool loc_site_ions_hash(const LocCtx &c, const HashLds &h, int S) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const DevConfig *cfg = c.cfg;
    const int T = cfg->n_types, Lm1 = c.L - 1, zmax = c.zmax;
    const bool nn = c.nl.n_nl != 0;
    const uint64_t types64 = load_types64(cfg);
    const float err = cfg->mz_error;
    if (c.b->debug & 8192u) return true;
    /* depth of the largest score gap (Ascore.cpp:164-172) */
    if (lane >= 1 && lane < S) {
        float best = 0.f;
        int depth = 0;
        for (int d = 0; d < PYA_NTOP; d++) {
            const float diff = w.scores[d] - w.scores[lane * 10 + d];
            if (diff > best) {
                best = diff;
                depth = d;
            }
        }
        w.c_depth[lane] = depth;
    }
    for (int i = lane; i < S * 2; i += 64) {
        w.c_cnt[i] = 0;
        w.c_tr[i] = 0;
    }
    if (Lm1 <= 0 || S < 2) {
        wave_lds_sync();
        return false;
    }
    STAMP_BEGIN();
    /* largest running sum (for the rounding allowance below) */
    const float tau = err * 0.125f;
    float runmax = 0.f;
    for (int i = lane; i < S * 2 * Lm1; i += 64) {
        const int sd = i / Lm1, p = i - sd * Lm1;
        runmax = __builtin_fmaxf(runmax, __builtin_fabsf(w.run[(size_t)sd * c.pos_cap + p]));
    }
    runmax = wave_max_f32(runmax);
    /* |twin - ion| <= tau + slop: the running sums differ by at most tau, the float32 loss subtraction and the
     * final narrowing round by half an ulp each on either side */
    const float slop = (runmax + 64.f) * 6e-7f;
    const float margin = tau + slop;
    const float reach = err + margin + slop;
    if (!(margin + slop < 0.5f * err)) return true;          /* (huge masses against a tiny tolerance) */
    const float qr = reach + slop;                           /* (the window of the alone-within-reach test) */
    /* (the closed forms of lh_resolve ask for nothing else within rf of an ion and its one neighbour) */
    const float rf = 4.f * err;
    const float qf = rf + slop;                              /* cells asked: those of x -+ qf */
    const float inv_cw = 1.f / (LH_CELL_QR * qr);           /* (wide cells: an ion's window lies in one cell most of the time, and cells stay almost empty) */
    const int hshift = 32 - (31 - __builtin_clz(h.hs));
    const float R = 8.f * err;                               /* window of the exact route */
    const FastDiv divZ = fastdiv_make((uint32_t)zmax);
    wave_lds_sync();
    STAMP_T(*c.b, 30, false);
    int staged = 0, lists_dir = -1;
    for (int t = 0; t < T; t++) {
        const int d = t < cfg->n_fwd ? 0 : 1;
        if (d != lists_dir) {
            wave_lds_sync();
            if (lh_build_lists(c, h, S, d, tau, nn)) return true;
            lists_dir = d;
            wave_lds_sync();
            STAMP_T(*c.b, 60, false);
        }
        double A, B;
        type_constants(type_at(types64, t), &A, &B);
        const int offW = (int)h.off[0], nW = (int)h.cnt[0] * zmax;
        if (nW > (int)h.vc) return true;
        /* competitors go together while the table holds them, else one at a time */
        int c0 = 1;
        while (c0 < S) {
            int c1 = c0, nB = 0;
            while (c1 < S) {
                const int n = (int)h.cnt[2 + 2 * (c1 - 1)] * zmax;
                if (nW + nB + n > (int)h.vc) break;
                nB += n;
                c1++;
            }
            if (c1 == c0) return true;                        /* one competitor's in-span ions do not fit */
            /* ---- table: values, then the grid ---- */
            for (int i = lane; i < (int)(h.hs / 2); i += 64) h.tab[i] = 0u;
            /* a lane per (prefix, variant) pair, the charges in a wave-uniform inner loop: the pair is decoded and its neutral
             * m/z formed once, and the charge arithmetic runs the one path its charge needs (lanes with their own charges
             * ran all of them) */
            for (int pair = lane; pair < (int)h.cnt[0]; pair += 64) {
                const uint32_t pz = h.pairs[offW + pair];
                const int p = (int)(pz & 255u), v = (int)(pz >> 8);
                const float run = w.run[(size_t)d * c.pos_cap + p];
                const float x = nn ? run - c.nl.uniq[v] : run;
                const double m = ((double)x + A) - B;
                float *dst = h.val + pair * zmax;
                for (int z = 1; z <= zmax; z++) dst[z - 1] = charge_mz(m, z);
            }
            {
                /* ... and the in-span pairs of all the group's competitors in one item space (a dozen pairs each) */
                int np_all = 0;
                for (int cc = c0; cc < c1; cc++) np_all += (int)h.cnt[2 + 2 * (cc - 1)];
                for (int pi = lane; pi < np_all; pi += 64) {
                    int cc = c0, first = 0;
                    {
                        int acc = 0;
                        for (int q2 = c0; q2 < c1; q2++) {
                            if (pi >= acc) {
                                cc = q2;
                                first = acc;
                            }
                            acc += (int)h.cnt[2 + 2 * (q2 - 1)];
                        }
                    }
                    const uint32_t pz = h.pairs[(int)h.off[2 + 2 * (cc - 1)] + (pi - first)];
                    const int p = (int)(pz & 255u), v = (int)(pz >> 8);
                    const float run = w.run[(size_t)(cc * 2 + d) * c.pos_cap + p];
                    const float x = nn ? run - c.nl.uniq[v] : run;
                    const double m = ((double)x + A) - B;
                    float *dst = h.val + nW + pi * zmax;               /* (the competitors' lists follow each other in this order) */
                    for (int z = 1; z <= zmax; z++) dst[z - 1] = charge_mz(m, z);
                }
            }
            wave_lds_sync();
            STAMP_T(*c.b, 61, false);
            for (int i = lane; i < nW + nB; i += 64) lh_insert(h, lh_cell(h.val[i], inv_cw), i, hshift);
            wave_lds_sync();
            STAMP_T(*c.b, 31, false);
            /* ---- every in-span ion: alone within reach? ---- */
            uint32_t nslow = 0;
            LhPass q;
            q.d = d;
            q.zmax = zmax;
            q.c0 = c0;
            q.c1 = c1;
            q.rf = rf;
            q.qf = qf;
            q.inv_cw = inv_cw;
            q.nW = nW;
            q.offW = offW;
            q.hshift = hshift;
            q.A = A;
            q.B = B;
            q.err = err;
            q.margin = margin;
            q.R = R;
            q.nn = nn;
            q.divZ = divZ;
            /* the in-span ions of ALL the group's competitors in one item space (a competitor has ~100: two rounds of 64 lanes
             * with the second one a third full when they go one by one): every lane finds its competitor from the running
             * totals -- at most PYA_LOC_SB_MAX - 1 of them, wave-uniform values */
            int n_items = 0;
            for (int cc = c0; cc < c1; cc++) n_items += ((int)h.cnt[1 + 2 * (cc - 1)] + (int)h.cnt[2 + 2 * (cc - 1)]) * zmax;
            for (int base = 0; base < n_items; base += 64) {
                const int i = base + lane;
                const bool on = i < n_items;
                int cc = c0, first = 0, b_lo = nW;             /* this lane's competitor, its first item, its B ions' first entry */
                {
                    int acc = 0, bl = nW;
                    for (int q2 = c0; q2 < c1; q2++) {
                        const int nA_q = (int)h.cnt[1 + 2 * (q2 - 1)] * zmax, nB_q = (int)h.cnt[2 + 2 * (q2 - 1)] * zmax;
                        if (i >= acc) {
                            cc = q2;
                            first = acc;
                            b_lo = bl;
                        }
                        acc += nA_q + nB_q;
                        bl += nB_q;
                    }
                }
                const int offA = (int)h.off[1 + 2 * (cc - 1)];
                const int nA = (int)h.cnt[1 + 2 * (cc - 1)] * zmax, nBc = (int)h.cnt[2 + 2 * (cc - 1)] * zmax;
                const int b_hi = b_lo + nBc, li = i - first;
                bool hit = false;
                int id = 0, side = 0;
                float x = 0.f;
                if (on) {
                    if (li < nA) {
                        const int pair = (int)fastdiv((uint32_t)li, divZ);
                        id = (int)h.pairs[offA + pair] * zmax + (li - pair * zmax);
                    } else {
                        side = 1;
                        id = b_lo + (li - nA);
                    }
                    x = h.val[id];
#ifdef PYA_STAMPS
                    if (c.b->stamps) atomicAdd(&c.b->stamps[62], 1ull);      /* (diagnostic build: in-span ions asked) */
#endif
                    const int k0 = lh_cell(x - qr, inv_cw), k1 = lh_cell(x + qr, inv_cw);
                    hit = lh_probe2(h, k0, k1, hshift, id, x, reach, nW, b_lo, b_hi);
                    if (c.b->debug & 16384u) hit = true;
                }
                const uint64_t hm = __ballot(hit);
                if (hit) {
                    const uint32_t slot = nslow + (uint32_t)mask_rank(hm);
                    if (slot < LH_HIT_CAP) h.slow[slot] = (uint32_t)id | ((uint32_t)cc << 16) | ((uint32_t)side << 24);
                }
                nslow += (uint32_t)__popcll(hm);
                lh_stage_push(c, on && !hit, x, (uint32_t)(cc * 2 + side), staged);
                /* (one call site: after the last round, or when the list could overflow with the next 64 items) */
                if (nslow && (base + 64 >= n_items || nslow + 64u > LH_HIT_CAP)) {
                    if (lh_resolve(c, h, q, nslow, staged)) return true;
                    nslow = 0;
                }
            }
            STAMP_T(*c.b, 32, false);
            STAMP_T(*c.b, 33, false);
            wave_lds_sync();
            c0 = c1;
        }
    }
    while ((staged & 0xffff) > 0) lh_stage_flush(c, staged);
    wave_lds_sync();
    if (lane < S * 2) {                                      /* (the packed counts of lh_stage_flush) */
        const uint32_t v = w.c_tr[lane];
        w.c_tr[lane] = v & 0xffffu;
        w.c_cnt[lane] = v >> 16;
    }
    wave_lds_sync();
    STAMP_T(*c.b, 34, false);
    return false;
}

#endif
