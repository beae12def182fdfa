/* fused_pack.hip.h -- score + localize for SEVERAL PSMs per wavefront.
 *
 * The fused kernel (fused_core.hip.h) gives every PSM a wavefront of its own.  With C(n,k) site
 * assignments and two directions 2 x C(n,k) of its 64 lanes walk (40 for a 3-of-6 PSM, 4 for a 1-of-2
 * one), and everything after the walk -- scores, winner, competitors, their fragment lists, pairing,
 * Ascores: half of the kernel's vector instructions -- runs on 3 to 24 lanes.  These kernels are bound
 * by vector-instruction issue (profiles/r03_valu_ceiling.md: 3.6 cycles per wave64 instruction of
 * this mix whatever the lane mask), so idle lanes are lost time.  Here a wavefront takes G PSMs
 * ("slots", G = 64 / n_cap of the launch, at most PACK_MAX_SLOTS):
 *
 *   stage    per slot, wave-cooperative as before (letters -> residue masses, retained peaks into a
 *            peak pool shared by the slots, m/z grid);
 *   walk     one (slot, direction, site assignment) walker per lane, 64 / n_cap (slot, direction)
 *            groups per pass: a 20-assignment PSM fills 60 lanes, three PSMs take two passes of
 *            L - 1 steps instead of three;
 *   after    lane = (slot, site assignment) for scores, winner (per-slot maximum through LDS),
 *            single-move competitors; lane = (slot, signature of the round, direction) for the
 *            competitors' fragment m/z; the site-determining-ion items of all slots' tasks laid
 *            end to end; lane = (slot, competitor) for the Ascores.
 *
 * Same arithmetic, same window test, same std::sort emulation (run per slot when a slot's best score
 * is tied), same pairing rule as fused_core.hip.h: results are bit-identical.  Scope: fragment charge
 * 1, mz_error <= 0.49, at most 255 fragments per assignment.  A slot this body cannot take is passed
 * on, nothing written: to the one-PSM-per-wavefront kernel when its retained peaks do not fit the
 * pool or a residue mass is at or below two tolerances (`over` list), to the general localize
 * instantiation when an ion has two partners or introsort runs out of depth (`redo` list, with the
 * scores, count records and grid score_signatures would have left).
 */
#ifndef PYA_FUSED_PACK_H
#define PYA_FUSED_PACK_H
#include "fused_core.hip.h"

#define PACK_MAX_SLOTS 8
#define PACK_ROUND 3                  /* competitors of a slot localised together (plus its winner) */
#define PACK_LOAD_CHUNKS 6            /* chunks of 128 retained peaks fetched before the first is stored */

/* slot flags */
#define PK_ACTIVE 1u                  /* the slot holds a PSM this wavefront scores */
#define PK_DECLINED 2u                /* ... which goes to the general localize instantiation */
#define PK_BADSTATUS 4u               /* the slot's PSM carries an error status: "no result" is all it gets */

struct PackLds {
    /* per slot (arrays of PACK_MAX_SLOTS) */
    uint32_t *psm, *flags, *peak_at, *best_i, *n_pushed, *kmax, *tie_n, *order_off;
    int32_t *L, *N, *k, *R, *last_cell, *n_aux;
    float *base, *inv_w, *nb;
    uint64_t *site_mask, *best_bits;
    int64_t *s0, *p0, *pep0, *a0;
    uint8_t *site_pos;       /* [G][64] residue index of the j-th modifiable residue */
    uint8_t *let;            /* [G][64] peptide letters */
    float2 *resd;            /* [G][pos_cap + 1] */
    uint16_t *grid;          /* [G][PYA_GRID_CELLS] */
    float *mass_l;           /* [32] */
    uint32_t *flag_l;        /* [32] */
    /* walk region */
    uint32_t *cnt;           /* [5][64] rank histogram columns of a pass */
    PeakEntry *peaks;        /* [pool_cap] */
    /* post region (over the walk region) */
    unsigned char *sort_raw;
    PushedEntry *pushed;     /* [G][push_cap] */
    unsigned long long *site_alt;   /* [G][kc] */
    uint32_t *site_max, *site_tie;  /* [G][kc] */
    float *asc_min;          /* [G][kc] */
    float *sc;               /* [G][1 + PACK_ROUND][10] */
    uint32_t *c_tr, *c_cnt;  /* [G * PACK_ROUND * ndir][2] */
    int32_t *c_depth;        /* [G][PACK_ROUND] */
    uint32_t *c_site;        /* [G][PACK_ROUND] */
    float *asc_l;            /* [G][PACK_ROUND] */
    uint32_t *t_lo, *t_off;  /* [G * PACK_ROUND * ndir + 1] */
    float *selm;             /* [G][(1 + PACK_ROUND) * ndir][pos_cap] fragment m/z of the round's signatures */
    /* kept */
    uint8_t *rkl;            /* [G][pos_cap][stride] */
    uint32_t *acc;           /* [5][64] rank counts per (slot, assignment) lane, two 16-bit fields a word */
    uint32_t *rec;           /* [64][3] cumulative counts as bytes */
    float *wsl;              /* [64] */
};

__host__ __device__ static inline size_t pack_post_bytes(uint32_t G, uint32_t n_cap, uint32_t push_cap, uint32_t kc, uint32_t ndir,
                                                         uint32_t pos_cap) {
    const size_t ntask = (size_t)G * PACK_ROUND * ndir;
    return fused_align16(fused_sort_bytes(n_cap)) + (size_t)G * push_cap * 16 + (size_t)G * kc * (8 + 4 + 4 + 4) +
           (size_t)G * (1 + PACK_ROUND) * 40 + ntask * 16 + (size_t)G * PACK_ROUND * 12 + fused_align16((ntask + 1) * 8) + 64 +
           (size_t)G * (1 + PACK_ROUND) * ndir * pos_cap * 4;            /* + the fragment m/z lists of a round */
}
__host__ __device__ static inline size_t pack_walk_bytes(uint32_t pool_cap) {
    return PYA_NTOP / 2 * 64 * 4 + (size_t)pool_cap * 8;
}
__host__ __device__ static inline size_t pack_fixed_bytes(uint32_t G, uint32_t pos_cap) {
    return (size_t)PACK_MAX_SLOTS * (17 * 4 + 6 * 8) + 64 + (size_t)G * 128 + fused_align16((size_t)G * (pos_cap + 1) * 8) +
           (size_t)G * PYA_GRID_CELLS * 2 + 256;
}
__host__ __device__ static inline size_t pack_kept_bytes(uint32_t G, uint32_t stride, uint32_t pos_cap, uint32_t ndir) {
    return fused_align16((size_t)G * pos_cap * stride) + PYA_NTOP / 2 * 64 * 4 + 64 * 12 + 64 * 4;
}
__host__ __device__ static inline size_t pack_lds_bytes(uint32_t G, uint32_t pool_cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
                                                        uint32_t push_cap, uint32_t kc, uint32_t ndir) {
    const size_t walk = pack_walk_bytes(pool_cap), post = pack_post_bytes(G, n_cap, push_cap, kc, ndir, pos_cap);
    return pack_fixed_bytes(G, pos_cap) + fused_align16(walk > post ? walk : post) + pack_kept_bytes(G, stride, pos_cap, ndir) + 32;
}

DEV PackLds pack_carve(unsigned char *raw, uint32_t G, uint32_t pool_cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap,
                       uint32_t push_cap, uint32_t kc, uint32_t ndir) {
    PackLds f;
    size_t o = 0;
    uint64_t *q = (uint64_t *)(raw + o);
    f.site_mask = q;
    f.best_bits = q + PACK_MAX_SLOTS;
    f.s0 = (int64_t *)(q + 2 * PACK_MAX_SLOTS);
    f.p0 = (int64_t *)(q + 3 * PACK_MAX_SLOTS);
    f.pep0 = (int64_t *)(q + 4 * PACK_MAX_SLOTS);
    f.a0 = (int64_t *)(q + 5 * PACK_MAX_SLOTS);
    o += 6 * 8 * PACK_MAX_SLOTS;
    uint32_t *w = (uint32_t *)(raw + o);
    f.psm = w;
    f.flags = w + 1 * PACK_MAX_SLOTS;
    f.peak_at = w + 2 * PACK_MAX_SLOTS;
    f.best_i = w + 3 * PACK_MAX_SLOTS;
    f.n_pushed = w + 4 * PACK_MAX_SLOTS;
    f.kmax = w + 5 * PACK_MAX_SLOTS;
    f.tie_n = w + 6 * PACK_MAX_SLOTS;
    f.order_off = w + 7 * PACK_MAX_SLOTS;
    f.L = (int32_t *)(w + 8 * PACK_MAX_SLOTS);
    f.N = (int32_t *)(w + 9 * PACK_MAX_SLOTS);
    f.k = (int32_t *)(w + 10 * PACK_MAX_SLOTS);
    f.R = (int32_t *)(w + 11 * PACK_MAX_SLOTS);
    f.last_cell = (int32_t *)(w + 12 * PACK_MAX_SLOTS);
    f.base = (float *)(w + 13 * PACK_MAX_SLOTS);
    f.inv_w = (float *)(w + 14 * PACK_MAX_SLOTS);
    f.nb = (float *)(w + 15 * PACK_MAX_SLOTS);
    f.n_aux = (int32_t *)(w + 16 * PACK_MAX_SLOTS);
    o += 17 * 4 * PACK_MAX_SLOTS;
    o = fused_align16(o) + 48;
    f.site_pos = raw + o;
    o += (size_t)G * 64;
    f.let = raw + o;
    o += (size_t)G * 64;
    f.resd = (float2 *)(raw + o);
    o += fused_align16((size_t)G * (pos_cap + 1) * 8);
    f.grid = (uint16_t *)(raw + o);
    o += (size_t)G * PYA_GRID_CELLS * 2;
    f.mass_l = (float *)(raw + o);
    f.flag_l = (uint32_t *)(raw + o + 128);
    o += 256;
    /* walk region */
    f.cnt = (uint32_t *)(raw + o);
    f.peaks = (PeakEntry *)(raw + o + PYA_NTOP / 2 * 64 * 4);
    /* post region over it */
    size_t p = o;
    f.sort_raw = raw + p;
    p += fused_align16(fused_sort_bytes(n_cap));
    f.pushed = (PushedEntry *)(raw + p);
    p += (size_t)G * push_cap * 16;
    f.site_alt = (unsigned long long *)(raw + p);
    p += (size_t)G * kc * 8;
    f.site_max = (uint32_t *)(raw + p);
    p += (size_t)G * kc * 4;
    f.site_tie = (uint32_t *)(raw + p);
    p += (size_t)G * kc * 4;
    f.asc_min = (float *)(raw + p);
    p += (size_t)G * kc * 4;
    f.sc = (float *)(raw + p);
    p += (size_t)G * (1 + PACK_ROUND) * 40;
    const size_t ntask = (size_t)G * PACK_ROUND * ndir;
    f.c_tr = (uint32_t *)(raw + p);
    p += ntask * 8;
    f.c_cnt = (uint32_t *)(raw + p);
    p += ntask * 8;
    f.c_depth = (int32_t *)(raw + p);
    p += (size_t)G * PACK_ROUND * 4;
    f.c_site = (uint32_t *)(raw + p);
    p += (size_t)G * PACK_ROUND * 4;
    f.asc_l = (float *)(raw + p);
    p += (size_t)G * PACK_ROUND * 4;
    f.t_lo = (uint32_t *)(raw + p);
    f.t_off = f.t_lo + ntask + 1;
    p += fused_align16((ntask + 1) * 8) + 64;
    f.selm = (float *)(raw + p);
    const size_t walk = pack_walk_bytes(pool_cap), post = pack_post_bytes(G, n_cap, push_cap, kc, ndir, pos_cap);
    o += fused_align16(walk > post ? walk : post);
    f.rkl = raw + o;
    o += fused_align16((size_t)G * pos_cap * stride);
    f.acc = (uint32_t *)(raw + o);
    o += PYA_NTOP / 2 * 64 * 4;
    f.rec = (uint32_t *)(raw + o);
    o += 64 * 12;
    f.wsl = (float *)(raw + o);
    return f;
}

/* residue mask of a site assignment with a per-lane position table (the wave-uniform deposit_sites
 * loops over the site mask on the scalar unit; lanes of different PSMs have different masks) */
DEV uint64_t deposit_sites_lane(uint64_t bits, const uint8_t *pos) {
    uint64_t out = 0;
    while (bits) {
        const int j = __builtin_ctzll(bits);
        bits &= bits - 1;
        out |= 1ull << pos[j];
    }
    return out;
}

/* BOTH: ion types of both directions.  pdesc: [n_ids][8] the PSMs' descriptors in launch order (six words of
 * BatchDev.desc, the PSM id, a spare), `first`: index of this wavefront's first PSM in it. */
template <bool BOTH>
DEV void fused_pack_body(const BatchDev &b, const uint64_t *pdesc, uint32_t n_ids, uint32_t first, uint32_t G, unsigned char *lds_raw,
                         uint32_t pool_cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t push_cap, uint32_t kc,
                         uint32_t *redo_count, uint32_t *redo_ids, uint32_t *over_count, uint32_t *over_ids) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const int ndir = BOTH ? 2 : 1;
    const PackLds f = pack_carve(lds_raw, G, pool_cap, n_cap, stride, pos_cap, push_cap, kc, (uint32_t)ndir);
    const uint32_t max_k = b.max_k;
    const float err = cfg->mz_error;
    const int fixed_dir = cfg->n_fwd > 0 ? 0 : 1;
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;              /* ion-type offsets per direction of travel */
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const float wide_min = 2.f * cfg->mz_error + 0.02f;

    STAMP_BEGIN();
    if (lane < 32) {
        f.mass_l[lane] = cfg->res_mass[lane];
        f.flag_l[lane] = cfg->res_modifiable[lane];
    }
    if (lane < PACK_MAX_SLOTS) {
        f.flags[lane] = 0u;
        f.n_pushed[lane] = 0u;
        f.kmax[lane] = 0u;
        f.L[lane] = 1;
        f.N[lane] = 0;
        f.k[lane] = 0;
        f.R[lane] = 0;
        f.peak_at[lane] = 0u;
        f.order_off[lane] = 0u;
    }
#pragma unroll
    for (int d = 0; d < PYA_NTOP / 2; d++) f.acc[d * 64 + lane] = 0u;
    wave_lds_sync();

    /* ---------------- stage ----------------
     * A wavefront's time here is memory round trips, and it has few neighbours on its CU to hide them
     * (LDS decides the occupancy), so nothing is fetched slot after slot: (1) one lane per slot fetches the
     * slot's PSM id, then its descriptor, status and retained-peak count; (2) the letters and the retained
     * peaks of ALL slots are fetched as flat lists, several loads per lane in flight; (3) only then the
     * per-slot arithmetic (residue masses, site masks, m/z grids) runs, on LDS alone. */
    {
        /* round trip 1: the slot's descriptor in launch order (host-packed: the six descriptor words + the PSM id) */
        const uint32_t gi = first + (uint32_t)lane;
        const bool have = (uint32_t)lane < G && gi < n_ids;
        uint64_t dw0 = 0, dw1 = 0, dw2 = 0, dw3 = 0, dw4 = 0, dw5 = 0, dw6 = 0;
        if (have) {
            const uint64_t *dw = pdesc + (size_t)gi * 8;
            dw0 = dw[0]; dw1 = dw[1]; dw2 = dw[2]; dw3 = dw[3]; dw4 = dw[4]; dw5 = dw[5]; dw6 = dw[6];
        }
        const uint32_t psm = (uint32_t)dw6;
        /* round trip 2: status and retained-peak count (slot lanes) and, beside them, the letters of every slot */
        int status = PYA_ST_OK, R = 0;
        if (have) {
            status = b.status[psm];
            R = (int)b.ret_n[psm];
        }
        uint32_t letv[PACK_MAX_SLOTS];
#pragma unroll
        for (int s = 0; s < PACK_MAX_SLOTS; s++) {
            letv[s] = (uint32_t)'A';
            if ((uint32_t)s < G) {                           /* (wave-uniform) */
                const uint32_t plo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)dw1, s);
                const uint32_t phi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(dw1 >> 32), s);
                const int Ls = __builtin_amdgcn_readlane((int)(uint32_t)dw4, s) & 0xffff;
                const int hv = __builtin_amdgcn_readlane((int)have, s);
                if (hv && lane < Ls) letv[s] = (uint32_t)b.pep[(int64_t)(((uint64_t)phi << 32) | plo) + lane];
            }
        }
        const bool okst = have && status == PYA_ST_OK;
        const uint32_t need = okst ? (((uint32_t)R + PYA_TABLE_PAD + 1u) & ~1u) : 0u;     /* even: the lookup reads 16-byte pairs */
        int total_need;
        const uint32_t at = (uint32_t)wave_excl_scan_i32((int)need, &total_need);
        const bool fits = okst && at + need <= pool_cap && !(b.debug & 0x8000u);
        if (have && !okst) {
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        if (okst && !fits) over_ids[atomicAdd(over_count, 1u)] = psm;    /* the one-PSM-per-wavefront kernel takes it */
        if ((uint32_t)lane < G) {
            f.psm[lane] = psm;
            f.flags[lane] = fits ? (PK_ACTIVE | ((b.debug & 512u) ? PK_DECLINED : 0u)) : (have && !okst ? PK_BADSTATUS : 0u);
            f.peak_at[lane] = fits ? at : 0u;
            f.order_off[lane] = (uint32_t)(dw5 >> 32);
            f.L[lane] = fits ? (int)(dw4 & 0xffffu) : 1;
            f.n_aux[lane] = (int)((dw4 >> 16) & 0xffffu);
            f.k[lane] = (int)((dw4 >> 32) & 0xffffu);
            f.N[lane] = fits ? (int)(uint32_t)dw5 : 0;
            f.R[lane] = fits ? R : 0;
            f.p0[lane] = (int64_t)dw0;
            f.pep0[lane] = (int64_t)dw1;
            f.s0[lane] = (int64_t)dw2;
            f.a0[lane] = (int64_t)dw3;
        }
#pragma unroll
        for (int s = 0; s < PACK_MAX_SLOTS; s++)
            if ((uint32_t)s < G) f.let[s * 64 + lane] = (uint8_t)letv[s];
    }
    wave_lds_sync();
    STAMP_T(b, 60, );
    /* result rows of every slot that has a PSM start out empty (also the ones with an error status) */
    for (uint32_t s = 0; s < G; s++) {
        if (!(f.flags[s] & (PK_ACTIVE | PK_BADSTATUS))) continue;
        const uint32_t psm = f.psm[s];
        for (uint32_t a = lane; a < max_k; a += 64) {
            b.ascores[(size_t)psm * max_k + a] = 0.f;
            b.alt_mask[(size_t)psm * max_k + a] = 0ull;
        }
    }
    STAMP_T(b, 61, );
    /* retained peaks of all slots, laid end to end in the pool: round trip 3, two entries (16 bytes) per lane and
     * load, up to PACK_LOAD_CHUNKS x 128 entries in flight per wavefront (LDS, not registers, limits this kernel's
     * occupancy: the loads can have them).  Slot tables start at even pool offsets and even workspace offsets. */
    {
        uint32_t pool_used = 0;
        for (uint32_t s = 0; s < G; s++)
            if (f.flags[s] & PK_ACTIVE) pool_used = f.peak_at[s] + (((uint32_t)f.R[s] + PYA_TABLE_PAD + 1u) & ~1u);
        const uint32_t pairs = pool_used >> 1;
        const uint32_t inf = __float_as_uint(__builtin_huge_valf());
        uint4 *d4 = (uint4 *)f.peaks;
        for (uint32_t base = 0; base < pairs; base += 64 * PACK_LOAD_CHUNKS) {
            uint4 v[PACK_LOAD_CHUNKS];
#pragma unroll
            for (int q = 0; q < PACK_LOAD_CHUNKS; q++) {
                const uint32_t jp = base + q * 64 + (uint32_t)lane;    /* pair index in the pool */
                const uint32_t j = 2u * jp;
                uint32_t s = 0;                              /* the slot whose table holds entry j */
                for (uint32_t t = 1; t < G; t++)
                    if ((f.flags[t] & PK_ACTIVE) && j >= f.peak_at[t]) s = t;
                const uint32_t i = j - f.peak_at[s];
                const uint32_t Rs = (f.flags[s] & PK_ACTIVE) ? (uint32_t)f.R[s] : 0u;
                v[q] = make_uint4(inf, PYA_NO_MATCH, inf, PYA_NO_MATCH);   /* (beyond a slot's peaks: its +inf sentinels) */
                if (jp < pairs && i < Rs) {
                    v[q] = ((const uint4 *)(b.ret + f.p0[s]))[i >> 1];
                    if (i + 1 >= Rs) {
                        v[q].z = inf;
                        v[q].w = PYA_NO_MATCH;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < PACK_LOAD_CHUNKS; q++) {
                const uint32_t jp = base + q * 64 + (uint32_t)lane;
                if (jp < pairs) d4[jp] = v[q];
            }
        }
    }
    wave_lds_sync();
    STAMP_T(b, 62, );
    /* per slot, on LDS: residues (ModifiedPeptide.cpp:24-79), site mask, grid */
    for (uint32_t s = 0; s < G; s++) {
        if (!(f.flags[s] & PK_ACTIVE)) continue;
        const int L = f.L[s], n_aux = f.n_aux[s], R = f.R[s];
        const int64_t a0 = f.a0[s];
        const uint32_t letter = f.let[s * 64 + lane];
        const bool in = lane < L;
        const uint32_t li = (letter - 'A') & 31u;
        float m0 = f.mass_l[li];
        const bool modifiable = in && ((f.flag_l[li] & 1u) || (cfg->allow_n && lane == 0) || (cfg->allow_c && lane == L - 1));
        float m1 = m0 + cfg->mod_mass;
        for (int base = 0; base < n_aux; base += 64) {       /* fixed modifications (rare here: fetched per slot) */
            const int n = n_aux - base < 64 ? n_aux - base : 64;
            uint32_t aux_pos = 0;
            float aux_mass = 0.f;
            if (lane < n) {
                aux_pos = b.aux_pos[a0 + base + lane];
                aux_mass = b.aux_mass[a0 + base + lane];
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
        const uint64_t site_mask = __ballot(modifiable);
        /* every residue heavier than two tolerances (fused_core.hip.h: `wide`, which implies ascending lists) */
        const bool wide = !__any(in && !(m0 > wide_min && m1 > wide_min));
        if (!wide) {
            /* not this kernel's PSM: the one-PSM-per-wavefront kernel takes it (its result rows stay empty) */
            if (lane == 0) {
                over_ids[atomicAdd(over_count, 1u)] = f.psm[s];
                f.flags[s] = 0u;
                f.N[s] = 0;
                f.L[s] = 1;
            }
            wave_lds_sync();
            continue;
        }
        if (in) f.resd[(size_t)s * (pos_cap + 1) + lane] = make_float2(m0, m1);
        if (modifiable) f.site_pos[s * 64 + __popcll(site_mask & lanemask_lt())] = (uint8_t)lane;
        PeakTable tab;
        tab.e = f.peaks + f.peak_at[s];
        tab.n = R;
        tab.err = err;
        tab.half_check = false;
        grid_build(&tab, f.grid + (size_t)s * PYA_GRID_CELLS);
        if (lane == 0) {
            f.last_cell[s] = tab.last_cell;
            f.base[s] = tab.base;
            f.inv_w[s] = tab.inv_w;
            f.nb[s] = tab.nb;
            f.site_mask[s] = site_mask;
        }
        wave_lds_sync();
    }
    STAMP_T(b, 51, );
    /* a slot of this wavefront that holds a PSM: lanes without a walker run the walk's code on its tables
     * (the lookup of a lane must always see a table that ends in sentinels) */
    uint32_t any_slot = PACK_MAX_SLOTS;
    for (uint32_t s = 0; s < G; s++)
        if ((f.flags[s] & PK_ACTIVE) && any_slot == PACK_MAX_SLOTS) any_slot = s;
    if (any_slot == PACK_MAX_SLOTS) return;                  /* nothing to score here */

    /* ---------------- walk: one (slot, direction, assignment) walker per lane ---------------- */
    const FastDiv divN = fastdiv_make(n_cap);
    const uint32_t gpp = 64u / n_cap;                        /* (slot, direction) groups per pass */
    const uint32_t groups = G * (uint32_t)ndir;
    for (uint32_t g0 = 0; g0 < groups; g0 += gpp) {
        const uint32_t gl = fastdiv((uint32_t)lane, divN);
        const int sig = lane - (int)(gl * n_cap);
        const uint32_t g = g0 + gl;
        const bool inr = gl < gpp && g < groups;
        const uint32_t slot = inr ? (BOTH ? g >> 1 : g) : 0u;
        const int dir = BOTH ? (int)(g & 1u) : fixed_dir;
        const int L = f.L[slot], N = f.N[slot];
        const bool active = inr && (f.flags[slot] & PK_ACTIVE) && sig < N;
        const uint64_t bits = active ? b.order_tab[f.order_off[slot] + (uint32_t)sig] : 0ull;
        const uint64_t resmask = deposit_sites_lane(bits, f.site_pos + slot * 64);
        const uint32_t ts = (inr && (f.flags[slot] & PK_ACTIVE)) ? slot : any_slot;     /* whose tables the lane looks at */
        PeakTable tab;
        tab.e = f.peaks + f.peak_at[ts];
        tab.cell = f.grid + (size_t)ts * PYA_GRID_CELLS;
        tab.n = f.R[ts];
        tab.err = err;
        tab.half_check = false;
        tab.base = f.base[ts];
        tab.inv_w = f.inv_w[ts];
        tab.nb = f.nb[ts];
        tab.last_cell = f.last_cell[ts];
#pragma unroll
        for (int d = 0; d < PYA_NTOP / 2; d++) f.cnt[d * 64 + lane] = 0u;
        const int Lm1 = L - 1;
        const int steps = (int)wave_max_u32(active ? (uint32_t)Lm1 : 0u);
        const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
        const uint64_t M = msb_first_from(tmask, 0);
        const float2 *rp = f.resd + (size_t)slot * (pos_cap + 1) + (dir ? L - 1 : 0);
        const int rstride = dir ? -1 : 1;
        uint32_t *col = f.cnt + lane;
        uint8_t *ro = f.rkl + (size_t)slot * pos_cap * stride + (active ? (uint32_t)((BOTH ? dir * N : 0) + sig) : stride - 1u);
        const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
        float running = 0.f;
        wave_lds_sync();
        for (int seg = 0; seg < 2; seg++) {                  /* the mask words change after 32 steps */
            StepBits sb = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
            int c = (steps < seg * 32 + 32 ? steps : seg * 32 + 32) - seg * 32;
            for (int i = 0; i < c; i++, rp += rstride, ro += stride) {
                const bool on = active && seg * 32 + i < Lm1;
                const float2 mm = *rp;
                running = (sb.next() ? mm.y : mm.x) + running;           /* ModifiedPeptide.cpp:385-389 */
                const double m = ((double)running + A) - B;      /* (B = 0.0 for b / c / y: x - 0.0 is x) */
                const Look kq = look4(tab, (float)(m + 1.007825));
                int rk = kq.best;
                if (kq.more()) rk = look_rest(tab, kq);
                hist_bump(col, on, rk);
                if (on) *ro = (uint8_t)rk;
            }
        }
        wave_lds_sync();
        /* the pass's rank counts into the (slot, assignment) accumulators: both directions add up there */
        const uint32_t at = slot * n_cap + (uint32_t)sig;
#pragma unroll
        for (int d = 0; d < PYA_NTOP / 2; d++) {
            const uint32_t v = f.cnt[d * 64 + lane];
            if (active && v) atomicAdd(&f.acc[d * 64 + at], v);
        }
        wave_lds_sync();
    }

    STAMP_T(b, 52, );
    /* ---------------- lane = (slot, assignment) from here on ---------------- */
    const uint32_t slot = fastdiv((uint32_t)lane, divN) < G ? fastdiv((uint32_t)lane, divN) : 0u;
    const int sig = lane - (int)(fastdiv((uint32_t)lane, divN) * n_cap);
    const bool in_slots = fastdiv((uint32_t)lane, divN) < G;
    const int L = f.L[slot], N = f.N[slot], k = f.k[slot], Lm1 = L - 1;
    const uint32_t nfrag = (uint32_t)ndir * (uint32_t)Lm1;   /* <= 255 (host) */
    const bool sig_lane = in_slots && (f.flags[slot] & PK_ACTIVE) && sig < N;
    const uint64_t my_bits = sig_lane ? b.order_tab[f.order_off[slot] + (uint32_t)sig] : 0ull;
    const uint8_t *spos = f.site_pos + slot * 64;
    float my_ws = 0.f;
    int fail = 0;
    if (sig_lane) {
        /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
        uint32_t cum[PYA_NTOP];
        uint32_t run = 0;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d++) {
            run += (f.acc[(d >> 1) * 64 + lane] >> ((d & 1) * 16)) & 0xffffu;
            cum[d] = run;
        }
        my_ws = -1.f;
        if (nfrag <= b.lut_n_max) {
            double sum = 0.;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                const float sc = lut_score(b, (uint32_t)d, cum[d], nfrag);
                const float prod = cfg->weights[d] * sc;                  /* float product ...   */
                sum = sum + (double)prod;                                 /* ... double sum      */
            }
            my_ws = (float)sum;
        } else {
            fail = 1;
        }
        uint32_t *r3 = f.rec + (size_t)lane * 3;              /* counts <= 255: a byte each */
        r3[0] = cum[0] | cum[1] << 8 | cum[2] << 16 | cum[3] << 24;
        r3[1] = cum[4] | cum[5] << 8 | cum[6] << 16 | cum[7] << 24;
        r3[2] = cum[8] | cum[9] << 8;
        f.wsl[lane] = my_ws;
    }
    /* a slot whose trial count is outside the score table: status + "no result", out of the rest */
    {
        const uint64_t fm = __ballot(fail != 0);
        if (fm) {
            wave_lds_sync();
            if (fail && sig == 0) {
                const uint32_t psm = f.psm[slot];
                b.status[psm] = PYA_ST_LUT_RANGE;
                b.best_score[psm] = -1.f;
                b.best_sig[psm] = 0ull;
                b.n_sig_out[psm] = -1;
                f.flags[slot] = 0u;
            }
            wave_lds_sync();
        }
    }
    wave_lds_sync();                                        /* cnt / peaks are free from here on */
    const bool live = in_slots && (f.flags[slot] & PK_ACTIVE) != 0u;
    const bool slane = live && sig < N;                     /* (slot, assignment) lane of a live slot */

    STAMP_T(b, 53, );
    /* ---- winner: the front of std::sort (cpp/Ascore.cpp:141-146) ---- */
    const uint32_t u = slane ? __float_as_uint(my_ws) : 0u;  /* scores are >= 0: bit order = value order */
    if (slane) atomicMax(&f.kmax[slot], u);
    for (uint32_t i = lane; i < G * kc; i += 64) {
        f.site_max[i] = 0u;
        f.site_tie[i] = 0u;
        f.site_alt[i] = 0ull;
        f.asc_min[i] = __builtin_huge_valf();
    }
    wave_lds_sync();
    const uint32_t kmax = f.kmax[slot];
    {
        const uint64_t at_max = __ballot(slane && u == kmax);
        const uint32_t sh = slot * n_cap;
        const uint64_t mine = (at_max >> sh) & (n_cap >= 64 ? ~0ull : ((1ull << n_cap) - 1ull));
        if (live && sig == 0) {
            f.tie_n[slot] = (uint32_t)__popcll(mine);
            f.best_i[slot] = mine ? (uint32_t)__builtin_ctzll(mine) : 0u;
        }
    }
    wave_lds_sync();
    for (uint32_t s = 0; s < G; s++) {                       /* a tie for the best score: std::sort decides (wave-uniform per slot) */
        if (!(f.flags[s] & PK_ACTIVE) || (f.flags[s] & PK_DECLINED)) continue;
        if (f.tie_n[s] == 1u && !(b.debug & 1024u)) continue;
        const int Ns = f.N[s];
        const SortLds srt = sort_carve(f.sort_raw, Ns);
        if (lane < Ns) {
            srt.key[lane] = f.wsl[s * n_cap + lane];
            srt.idx[lane] = (uint16_t)lane;
        }
        wave_lds_sync();
        bool out_of_depth = false;
        if (!(b.debug & 8u)) out_of_depth = sort_introsort_loop<true, true>(srt, Ns, true);
        const uint64_t m2 = __ballot(lane < Ns && __float_as_uint(srt.key[lane]) == f.kmax[s]);
        if (lane == 0) {
            f.best_i[s] = srt.idx[__builtin_ctzll(m2)];
            if (out_of_depth) f.flags[s] |= PK_DECLINED;
        }
        wave_lds_sync();
    }
    const uint32_t best_i = f.best_i[slot];
    const float best_ws = __uint_as_float(kmax);
    if (slane && (uint32_t)sig == best_i) f.best_bits[slot] = my_bits;
    wave_lds_sync();
    const uint64_t best_bits = f.best_bits[slot];
    bool declined = (f.flags[slot] & PK_DECLINED) != 0u;

    STAMP_T(b, 54, );
    /* ---- single-move competitors (cpp/Ascore.cpp:212-254) ---- */
    {
        const uint64_t gone = best_bits & ~my_bits, came = my_bits & ~best_bits;
        const bool single = slane && !declined && __popcll(gone) == 1 && __popcll(came) == 1;
        const int a = single ? __popcll(best_bits & (gone - 1)) : 0;
        if (single) atomicMax(&f.site_max[slot * kc + a], u);
        wave_lds_sync();
        if (single && u == f.site_max[slot * kc + a]) {
            if ((double)__builtin_fabsf(best_ws - my_ws) < 1e-6) {
                /* ties the winner: Ascore 0 (Ascore.cpp:159-161), no ion work needed */
                f.site_tie[slot * kc + a] = 1u;
                atomicOr(&f.site_alt[slot * kc + a], 1ull << spos[__builtin_ctzll(came)]);
            } else {
                const uint32_t at = atomicAdd(&f.n_pushed[slot], 1u);
                if (at < push_cap) {
                    PushedEntry pe;
                    pe.bits = my_bits;
                    pe.ws = my_ws;
                    pe.idx = (uint32_t)sig;
                    f.pushed[(size_t)slot * push_cap + at] = pe;
                }
            }
        }
        wave_lds_sync();
    }
    uint32_t np_max = 0;
    {
        uint32_t np = live && !declined ? f.n_pushed[slot] : 0u;
        if (np > push_cap) np = push_cap;                   /* cannot happen: push_cap >= k * (n_sites - k) */
        if (b.debug & 16u) np = 0;
        np_max = wave_max_u32(np);
    }

    STAMP_T(b, 55, );
    /* ---- Ascores, PACK_ROUND competitors of every slot at a time ---- */
    const uint32_t per_slot_lists = (1u + PACK_ROUND) * (uint32_t)ndir;
    const FastDiv divLists = fastdiv_make(per_slot_lists), div40 = fastdiv_make((1u + PACK_ROUND) * 10u), div10 = fastdiv_make(10u);
    int p2 = 1;
    while (p2 <= (int)pos_cap) p2 <<= 1;                     /* the partner search covers indices 0 .. L - 1 of any slot */
    for (uint32_t e0 = 0; e0 < np_max; e0 += PACK_ROUND) {
        /* fragment m/z of the winner (first round only) and of the round's competitors: one lane per
         * (slot, signature of the round, direction), the walk without its lookups */
        {
            const uint32_t s = fastdiv((uint32_t)lane, divLists);
            const uint32_t r = (uint32_t)lane - s * per_slot_lists;
            const uint32_t sg = BOTH ? r >> 1 : r;
            const int dd = BOTH ? (int)(r & 1u) : fixed_dir;
            bool on = s < G;
            const uint32_t ss = on ? s : 0u;
            const uint32_t fl = f.flags[ss];
            uint32_t np = f.n_pushed[ss];
            if (np > push_cap) np = push_cap;
            on = on && (fl & PK_ACTIVE) && !(fl & PK_DECLINED) && e0 < np && (sg == 0 ? e0 == 0 : e0 + sg - 1 < np);
            const int Ls = f.L[ss];
            const uint64_t sbits = sg == 0 ? f.best_bits[ss] : f.pushed[(size_t)ss * push_cap + (on ? e0 + sg - 1 : 0u)].bits;
            const uint64_t rm = deposit_sites_lane(on ? sbits : 0ull, f.site_pos + ss * 64);
            const uint64_t tmask = dd ? (__brevll(rm) >> (64 - Ls)) : rm;
            const uint64_t M = msb_first_from(tmask, 0);
            const float2 *rp = f.resd + (size_t)ss * (pos_cap + 1) + (dd ? Ls - 1 : 0);
            const int rstride = dd ? -1 : 1;
            const double A = dd ? Ab : Af, B = dd ? Bb : Bf;
            float *out = f.selm + ((size_t)ss * per_slot_lists + r) * pos_cap;
            const int steps = (int)wave_max_u32(on ? (uint32_t)(Ls - 1) : 0u);
            float running = 0.f;
            int step = 0;
            for (int seg = 0; seg < 2; seg++) {
                StepBits sb = {seg ? (uint32_t)M : (uint32_t)(M >> 32)};
                const int end = steps < seg * 32 + 32 ? steps : seg * 32 + 32;
                for (; step < end; step++, rp += rstride) {
                    const float2 mm = *rp;
                    running = (sb.next() ? mm.y : mm.x) + running;
                    const double m = ((double)running + A) - B;
                    if (on && step < Ls - 1) out[step] = (float)(m + 1.007825);
                }
            }
        }
        STAMP_T(b, 56, );
        /* depth scores of the winner and the competitors, read off the score table */
        for (uint32_t i = (uint32_t)lane; i < G * (1u + PACK_ROUND) * 10u; i += 64) {
            const uint32_t s = fastdiv(i, div40), r = i - s * (1u + PACK_ROUND) * 10u;
            const uint32_t sg = fastdiv(r, div10), d = r - sg * 10u;
            const uint32_t fl = f.flags[s];
            uint32_t np = f.n_pushed[s];
            if (np > push_cap) np = push_cap;
            const bool on = (fl & PK_ACTIVE) && !(fl & PK_DECLINED) && e0 < np && (sg == 0 ? e0 == 0 : e0 + sg - 1 < np);
            if (on) {
                const uint32_t who = sg == 0 ? f.best_i[s] : f.pushed[(size_t)s * push_cap + e0 + sg - 1].idx;
                const uint32_t cum = (f.rec[(size_t)(s * n_cap + who) * 3 + (d >> 2)] >> ((d & 3) * 8)) & 0xffu;
                const uint32_t nf = (uint32_t)ndir * (uint32_t)(f.L[s] - 1);
                f.sc[(size_t)s * (1 + PACK_ROUND) * 10 + r] = b.lut[lut_row(nf) + d * (nf + 1) + cum];
            }
        }
        wave_lds_sync();
        /* one lane per (slot, competitor): moved site, alternative position, depth of the largest score gap */
        const uint32_t ntask = G * PACK_ROUND * (uint32_t)ndir;
        {
            const uint32_t s = (uint32_t)lane / PACK_ROUND, c = (uint32_t)lane - s * PACK_ROUND;
            const bool okc = s < G;
            const uint32_t ss = okc ? s : 0u;
            const uint32_t fl = f.flags[ss];
            uint32_t np = f.n_pushed[ss];
            if (np > push_cap) np = push_cap;
            const bool on = okc && (fl & PK_ACTIVE) && !(fl & PK_DECLINED) && e0 + c < np;
            if (on) {
                const PushedEntry pe = f.pushed[(size_t)ss * push_cap + e0 + c];
                const uint64_t bb = f.best_bits[ss];
                const uint64_t gone = bb & ~pe.bits, came = pe.bits & ~bb;
                const int a = __popcll(bb & (gone - 1));
                atomicOr(&f.site_alt[ss * kc + a], 1ull << f.site_pos[ss * 64 + __builtin_ctzll(came)]);
                f.c_site[lane] = (uint32_t)a;
                float best = 0.f;                           /* depth of the largest score gap (Ascore.cpp:164-172) */
                int depth = 0;
                const float *sw = f.sc + (size_t)ss * (1 + PACK_ROUND) * 10, *sc_c = sw + (c + 1) * 10;
                for (int d = 0; d < PYA_NTOP; d++) {
                    const float diff = sw[d] - sc_c[d];
                    if (diff > best) {
                        best = diff;
                        depth = d;
                    }
                }
                f.c_depth[lane] = depth;
            }
        }
        for (uint32_t i = (uint32_t)lane; i < ntask * 2; i += 64) {
            f.c_tr[i] = 0;
            f.c_cnt[i] = 0;
        }
        wave_lds_sync();
        STAMP_T(b, 57, );
        /* ---- site-determining ions (cpp/ModifiedPeptide.cpp:259-320), fused_core.hip.h's span form: only
         * the steps between the first and the last residue in which winner and competitor differ are
         * examined (every slot here is `wide`); task = (slot, competitor, direction), the items of all
         * tasks laid end to end ---- */
        uint32_t my_len = 0;
        if ((uint32_t)lane < ntask) {
            const uint32_t t = (uint32_t)lane;
            const uint32_t sc_ = BOTH ? t >> 1 : t;               /* slot * PACK_ROUND + competitor */
            const uint32_t s = sc_ / PACK_ROUND, c = sc_ - s * PACK_ROUND;
            const int dd = BOTH ? (int)(t & 1u) : fixed_dir;
            const uint32_t fl = f.flags[s];
            uint32_t np = f.n_pushed[s];
            if (np > push_cap) np = push_cap;
            const bool on = (fl & PK_ACTIVE) && !(fl & PK_DECLINED) && e0 + c < np;
            uint32_t lo = 0;
            if (on) {
                const uint64_t diff = f.best_bits[s] ^ f.pushed[(size_t)s * push_cap + e0 + c].bits;    /* site indices */
                const int r_lo = f.site_pos[s * 64 + __builtin_ctzll(diff)];
                const int r_hi = f.site_pos[s * 64 + 63 - __builtin_clzll(diff)];
                const int Ls = f.L[s];
                lo = (uint32_t)(dd ? Ls - 1 - r_hi : r_lo);
                my_len = (uint32_t)(2 * (r_hi - r_lo));           /* both sides */
            }
            f.t_lo[t] = lo;
        }
        int items = 0;
        {
            const int off = wave_excl_scan_i32((int)my_len, &items);
            if ((uint32_t)lane < ntask) f.t_off[lane] = (uint32_t)off;
            if (lane == 0) f.t_off[ntask] = (uint32_t)items;
        }
        wave_lds_sync();
        if (!(b.debug & 1u))
        for (int base = 0; base < items; base += 64) {
            const int e = base + lane;
            if (e < items) {
                /* the task this item belongs to: the last one that starts at or before it */
                uint32_t lo_t = 0, hi_t = ntask;
                while (hi_t - lo_t > 1) {
                    const uint32_t mid = (lo_t + hi_t) >> 1;
                    if ((uint32_t)e >= f.t_off[mid]) lo_t = mid;
                    else hi_t = mid;
                }
                /* (tasks without items share their offset with the next one: step past them) */
                uint32_t task = lo_t;
                while (f.t_off[task + 1] <= (uint32_t)e) task++;
                const int rem = e - (int)f.t_off[task];
                const int len = (int)(f.t_off[task + 1] - f.t_off[task]) >> 1;
                const int side = rem >= len ? 1 : 0;
                const int i = (int)f.t_lo[task] + rem - side * len;
                const uint32_t sc_ = BOTH ? task >> 1 : task;
                const uint32_t s = sc_ / PACK_ROUND, c = sc_ - s * PACK_ROUND;
                const int d = BOTH ? (int)(task & 1u) : 0;
                const int Ls1 = f.L[s] - 1, Ns = f.N[s];
                const float *la = f.selm + ((size_t)s * per_slot_lists + d) * pos_cap;                       /* winner     */
                const float *lb = f.selm + ((size_t)s * per_slot_lists + (1 + c) * ndir + d) * pos_cap;      /* competitor */
                const float *mine = side ? lb : la, *other = side ? la : lb;
                const float me = mine[i];
                int total;
                {
                    float df[4];
                    bool ok[4], sk[4];
#pragma unroll
                    for (int uu = 0; uu < 4; uu++) {
                        const int q = i - 1 + uu;
                        ok[uu] = q >= 0 && q < Ls1;
                        const float o = ok[uu] ? other[q] : (q < 0 ? -__builtin_huge_valf() : __builtin_huge_valf());
                        df[uu] = side ? (o - me) : (me - o);   /* always (winner's ion) - (competitor's ion) */
                        sk[uu] = side ? (df[uu] <= -err) : (df[uu] >= err);
                    }
                    const int w1 = (ok[1] && __builtin_fabsf(df[1]) < err) ? 1 : 0;
                    const int w2 = (ok[2] && __builtin_fabsf(df[2]) < err) ? 1 : 0;
                    const int w3 = (ok[3] && __builtin_fabsf(df[3]) < err) ? 1 : 0;
                    int cnt = -1;
                    if (sk[0] && !sk[1]) cnt = w1 + w2;      /* first candidate = index i     */
                    else if (sk[1] && !sk[2]) cnt = w2 + w3; /* first candidate = index i + 1 */
                    if (cnt < 0) cnt = partners_in_run(other, Ls1, p2, me, side, err);
                    total = cnt;
                }
                if (total > 1 || (b.debug & 2048u)) {
                    atomicOr(&f.flags[s], PK_DECLINED);     /* two partners: the general kernel's serial walk decides */
                } else if (total == 0) {
                    const uint32_t who = side ? f.pushed[(size_t)s * push_cap + e0 + c].idx : f.best_i[s];
                    atomicAdd(&f.c_tr[task * 2 + side], 1u);
                    if ((int)f.rkl[((size_t)s * pos_cap + i) * stride + (d * Ns + (int)who)] <= f.c_depth[s * PACK_ROUND + c])
                        atomicAdd(&f.c_cnt[task * 2 + side], 1u);
                }
            }
        }
        wave_lds_sync();
        STAMP_T(b, 58, );
        /* ---- Ascores (cpp/Ascore.cpp:200-209, :239-251, :305-313) ---- */
        {
            const uint32_t s = (uint32_t)lane / PACK_ROUND, c = (uint32_t)lane - s * PACK_ROUND;
            const bool okc = s < G;
            const uint32_t ss = okc ? s : 0u;
            const uint32_t fl = f.flags[ss];
            uint32_t np = f.n_pushed[ss];
            if (np > push_cap) np = push_cap;
            const bool on = okc && (fl & PK_ACTIVE) && !(fl & PK_DECLINED) && e0 + c < np;
            float asc = 0.f;
            if (on) {
                uint32_t tr0 = 0, tr1 = 0, n0 = 0, n1 = 0;
                for (int d = 0; d < ndir; d++) {            /* a competitor's tasks: one per direction */
                    const uint32_t t = (ss * PACK_ROUND + c) * ndir + d;
                    tr0 += f.c_tr[t * 2];
                    tr1 += f.c_tr[t * 2 + 1];
                    n0 += f.c_cnt[t * 2];
                    n1 += f.c_cnt[t * 2 + 1];
                }
                const uint32_t depth = (uint32_t)f.c_depth[lane];
                if (tr0 > b.lut_n_max || tr1 > b.lut_n_max) {
                    fail = 1;
                    b.status[f.psm[ss]] = PYA_ST_LUT_RANGE;
                } else {
                    const float sc0 = b.lut[lut_row(tr0) + depth * (tr0 + 1) + n0];
                    const float sc1 = b.lut[lut_row(tr1) + depth * (tr1 + 1) + n1];
                    asc = sc0 - sc1;
                }
            }
            /* the smallest Ascore per modified site: the round's competitors of a slot one after the other */
            for (uint32_t cc = 0; cc < PACK_ROUND; cc++) {
                if (on && c == cc) {
                    float *dst = &f.asc_min[ss * kc + f.c_site[lane]];
                    *dst = asc < *dst ? asc : *dst;
                }
                wave_lds_sync();
            }
        }
    }
    wave_lds_sync();

    STAMP_T(b, 59, );
    /* ---------------- results / hand-over ---------------- */
    declined = live && (f.flags[slot] & PK_DECLINED) != 0u;
    if (declined) {
        /* leave what score_signatures would have left; the general localize instantiation redoes the PSM */
        const int64_t s0 = f.s0[slot];
        if (sig < N) {
            b.ws[s0 + sig] = my_ws;
            if (b.rec) {
                const uint32_t *r3 = f.rec + (size_t)lane * 3;
                uint32_t *dst = b.rec + (s0 + sig) * PYA_REC_WORDS;
                uint32_t cum[PYA_NTOP];
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) cum[d] = (r3[d >> 2] >> ((d & 3) * 8)) & 0xffu;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d += 2) dst[d >> 1] = cum[d] | (cum[d + 1] << 16);
                dst[5] = nfrag;
            }
        }
        if (sig == 0) {
            const uint32_t psm = f.psm[slot];
            b.ws_top[(size_t)psm * 4 + 1] = 0u;             /* no summary of the scores: localize scans them */
            redo_ids[atomicAdd(redo_count, 1u)] = psm;
        }
    } else if (live) {
        const uint32_t psm = f.psm[slot];
        if (sig < k && sig < (int)max_k) {
            float asc = f.asc_min[slot * kc + sig];
            if (f.site_tie[slot * kc + sig]) asc = 0.f < asc ? 0.f : asc;
            b.ascores[(size_t)psm * max_k + sig] = asc;
            b.alt_mask[(size_t)psm * max_k + sig] = f.site_alt[slot * kc + sig];
        }
        if (sig == 0) {
            b.best_score[psm] = best_ws;
            b.best_sig[psm] = best_bits;
            b.n_sig_out[psm] = N;
        }
    }
    for (uint32_t s = 0; s < G; s++) {                       /* the grids of the slots handed over (wave-cooperative copies) */
        const uint32_t fl = f.flags[s];
        if ((fl & PK_ACTIVE) && (fl & PK_DECLINED))
            ((uint64_t *)(b.grid + (size_t)f.psm[s] * PYA_GRID_CELLS))[lane] = ((const uint64_t *)(f.grid + (size_t)s * PYA_GRID_CELLS))[lane];
    }
}

#endif
