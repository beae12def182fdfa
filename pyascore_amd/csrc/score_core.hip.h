/* score_core.hip.h -- body of score_signatures for one PSM (one wavefront); see
 * score_signatures.hip for the notes.  Shared with tiny_batch.hip. */
#ifndef PYA_SCORE_CORE_H
#define PYA_SCORE_CORE_H
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
    uint64_t ha, hb, hc;        /* rank counts of the prefix, 16-bit fields */
};

/* the same for the straight-line walker (no neutral losses, charge 1, one ion type per direction):
 * a direction's prefix has at most 63 fragments, so the ten rank counts fit 8-bit fields that add
 * without unpacking, and the entry shrinks from 40 to 16 bytes -- this kernel's speed follows its
 * occupancy, and with 2 x 64 entries per wave LDS is what limits it */
struct PrefixCompact {
    float running;
    uint32_t hi;                /* ranks 8, 9 */
    uint64_t lo;                /* ranks 0..7 */
};

/* ---------------------------------------------------------------------------------------
 * Shared nodes of the assignment tree (general settings: neutral losses, several ion types or charges per
 * direction, where one (signature, direction, step) costs up to variants x types x charges lookups).
 * Two signatures that agree on the modifiable residues a fragment contains give it the same float32 running
 * sum and the same loss variants -- the same sequence of additions -- so the fragment's matches are looked up
 * once per distinct (step, pattern of the sites passed) NODE instead of once per signature (cfg4: 188 instead of
 * 380 per direction), which is how the reference shares prefixes (cpp/Ascore.cpp:69-109,
 * cpp/ModifiedPeptide.cpp:458-471).  Per direction:
 *   1. one lane per signature walks the residues (sums and loss state only); the lowest signature of every
 *      group -- signatures with the same pattern over the sites passed so far; host table per shape, behind
 *      the shape's order table -- writes the node (running sum, loss state, histogram column of its segment);
 *   2. one lane per node: every (variant, ion type, charge) lookup, ranks bumped into the column of the node's
 *      segment (level of sites passed x group);
 *   3. one lane per signature adds up the columns of its n_sites + 1 segments.
 * Shape table (64-bit words at order + N, N <= 64): own[dir][j] = signatures that are the lowest of their group
 * at level j; then bytes grp[dir][j][s] (rows padded to 8) = rank of the group of s among the groups of level j. */
struct NodeLds {
    const uint64_t *ntab;   /* LDS copy of the shape table */
    float *run;             /* [node_cap] running sum of the node's fragment */
    uint16_t *info;         /* [node_cap] loss state | histogram column << 8 */
    uint16_t *nb;           /* [64] first node of every step */
    uint16_t *sb;           /* [64] first column of every level */
    uint32_t *priv;         /* [PYA_NTOP / 2][64] a column per lane for the lookups of its node */
    const BatchDev *b;
    uint32_t node_cap, ncols;
};

/* one direction; false (nothing counted) when the PSM's nodes or columns do not fit */
DEV bool score_nodes_dir(const WalkEnv &e, const PeakTable &tab, const NodeLds &nd, uint64_t resmask, uint64_t site_mask,
                         int n_sites, int N, int dir, uint32_t acc[PYA_NTOP / 2], uint32_t &nfrag) {
    const int lane = lane_id();
    const int L = e.L, Lm1 = L - 1;
    const bool active = lane < N;
    const DevConfig *cfg = e.cfg;
    const int n_f = cfg->n_fwd, my_types = dir == 0 ? n_f : cfg->n_types - n_f, t_base = dir == 0 ? 0 : n_f;
    const int W8 = (N + 7) >> 3;
    const uint64_t *own = nd.ntab + dir * (n_sites + 1);
    const uint8_t *grp = (const uint8_t *)(nd.ntab + 2 * (n_sites + 1)) + (size_t)dir * (n_sites + 1) * W8 * 8;
    const uint64_t tsite = dir ? (__brevll(site_mask) >> (64 - L)) : site_mask;     /* sites in travel order */
    STAMP_BEGIN();
    int G = 0;
    if (lane < Lm1) G = __popcll(own[__popcll(tsite & ((2ull << lane) - 1ull))]);
    int nnodes, ncol;
    const int nbase = wave_excl_scan_i32(G, &nnodes);
    const int sbase = wave_excl_scan_i32(lane <= n_sites ? __popcll(own[lane]) : 0, &ncol);
    if ((uint32_t)nnodes > nd.node_cap || (uint32_t)ncol > nd.ncols || ncol > 255) return false;
    if (lane < Lm1) nd.nb[lane] = (uint16_t)nbase;
    if (lane <= n_sites) nd.sb[lane] = (uint16_t)sbase;
    for (uint32_t i = lane; i < (PYA_NTOP / 2) * nd.ncols; i += 64) e.cnt[i] = 0u;
#pragma unroll
    for (int d = 0; d < PYA_NTOP / 2; d++) nd.priv[d * 64 + lane] = 0u;
    wave_lds_sync();
    STAMP_T(*nd.b, 5, false);
    /* 1. walkers: sums and loss state; group owners write the nodes */
    {
        const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
        float running = 0.f;
        uint32_t nl_state = 0, g = 0, segb = 0;
        int j = 0;
        bool owner = active && ((own[0] >> lane) & 1ull);
        for (int step = 0; step < Lm1; step++) {
            const int ri = dir ? L - 1 - step : step;
            const float2 mm = e.resd[ri];
            const bool mod = (tmask >> step) & 1ull;
            running = (mod ? mm.y : mm.x) + running;                         /* ModifiedPeptide.cpp:385-389 */
            uint32_t nvar = 1u;
            if (e.n_nl) {
                const uint32_t nlp = e.resn[ri];
                const uint32_t cls = mod ? (nlp >> 4) : (nlp & 15u);
                if (cls) nl_state = nl_bump(nl_state, cls);
                nvar = (uint32_t)__popc((uint32_t)e.nl_present[nl_state & 255u]);
            }
            if ((tsite >> step) & 1ull) {                                    /* (wave-uniform) a site enters the fragment */
                j++;
                g = active ? (uint32_t)grp[(size_t)j * W8 * 8 + lane] : 0u;
                owner = active && ((own[j] >> lane) & 1ull);
                segb = nd.sb[j];
            }
            if (active) nfrag += nvar * (uint32_t)(my_types * e.zmax);
            if (owner) {
                const uint32_t at = (uint32_t)nd.nb[step] + g;
                nd.run[at] = running;
                nd.info[at] = (uint16_t)(nl_state | ((segb + g) << 8));
            }
        }
    }
    wave_lds_sync();
    STAMP_T(*nd.b, 6, false);
    /* 2. one lane per node.  The ranks of a node's lookups are bumped in the lane's own column (all lanes bumping
     * the column of a shared segment serialise on its address: 0.40 bank-conflict cycles per LDS cycle, measured)
     * and added to the segment's column once per round. */
    const uint64_t types64 = load_types64(cfg);
    uint32_t *mine = nd.priv + lane;
    for (int base = 0; base < nnodes; base += 64) {
        const int i = base + lane;
        const bool on_n = i < nnodes;
        const float running = on_n ? nd.run[i] : 0.f;
        const uint32_t inf = on_n ? (uint32_t)nd.info[i] : 0u;
        uint32_t pm = on_n ? (e.n_nl ? (uint32_t)e.nl_present[inf & 255u] : 1u) : 0u;
        while (__any(pm != 0)) {
            const bool on = pm != 0;
            const int v = on ? __builtin_ctz(pm) : 0;
            pm &= pm - 1;
            const float x = running - (e.n_nl ? e.nl_uniq[v] : 0.f);         /* float subtract (:572) */
            const double xd = (double)x;
            for (int t = 0; t < my_types; t++) {
                double A, B;
                type_constants(type_at(types64, t_base + t), &A, &B);
                const double m = (xd + A) - B;
                if (tab.half_check) {
                    for (int z = 1; z <= e.zmax; z++) hist_bump(mine, on, match_rank_lds(tab, charge_mz(m, z)));
                } else {
                    /* up to four charges of the fragment in flight: their lookups are independent chains */
                    for (int z0 = 1; z0 <= e.zmax; z0 += 4) {
                        const int nz = e.zmax - z0 + 1;                              /* (wave-uniform) */
                        const Look k0 = look4(tab, charge_mz(m, z0));
                        const Look k1 = look4(tab, charge_mz(m, nz > 1 ? z0 + 1 : z0));
                        const Look k2 = look4(tab, charge_mz(m, nz > 2 ? z0 + 2 : z0));
                        const Look k3 = look4(tab, charge_mz(m, nz > 3 ? z0 + 3 : z0));
                        int r0 = k0.best, r1 = k1.best, r2 = k2.best, r3 = k3.best;
                        if (k0.more()) r0 = look_rest(tab, k0);
                        if (k1.more()) r1 = look_rest(tab, k1);
                        if (k2.more()) r2 = look_rest(tab, k2);
                        if (k3.more()) r3 = look_rest(tab, k3);
                        hist_bump(mine, on, r0);
                        hist_bump(mine, on && nz > 1, r1);
                        hist_bump(mine, on && nz > 2, r2);
                        hist_bump(mine, on && nz > 3, r3);
                    }
                }
            }
        }
        uint32_t *col = e.cnt + (inf >> 8);
#pragma unroll
        for (int d = 0; d < PYA_NTOP / 2; d++) {
            const uint32_t c = mine[d * 64];
            mine[d * 64] = 0u;
            if (on_n && c) atomicAdd(col + (uint32_t)d * nd.ncols, c);
        }
    }
    wave_lds_sync();
    STAMP_T(*nd.b, 7, false);
    /* 3. a signature's counts = the columns of its segments */
    if (active) {
        for (int jj = 0; jj <= n_sites; jj++) {
            const uint32_t col = (uint32_t)nd.sb[jj] + (uint32_t)grp[(size_t)jj * W8 * 8 + lane];
#pragma unroll
            for (int d = 0; d < PYA_NTOP / 2; d++) acc[d] += e.cnt[(uint32_t)d * nd.ncols + col];
        }
    }
    wave_lds_sync();
    STAMP_T(*nd.b, 8, false);
    return true;
}

/* PREFIX is a template parameter so that the small-C(n,k) instantiation does not carry the
 * registers of the shared-prefix path (60 vs 77 VGPRs = 8 vs 6 waves per SIMD). */
/* NODES: the instantiation with the shared-node route (general settings); the others do not carry its registers */
template <bool PREFIX, bool NODES = false>
DEV void score_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t with_nl,
                    uint32_t compact, uint32_t node_cap = 0, uint32_t node_cols = 64, uint32_t node_words = 0,
                    uint32_t res_cap = 64, uint32_t nl_cap = 256) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;

    uint16_t *grid = (uint16_t *)lds_raw;                       /* [PYA_GRID_CELLS] */
    uint32_t *cnt = (uint32_t *)(grid + PYA_GRID_CELLS);        /* [PYA_NTOP / 2][64] */
    float2 *resd = (float2 *)(cnt + PYA_NTOP / 2 * (node_cap && node_cols > 64u ? node_cols : 64u));   /* [64] (the shared-node route has node_cols columns) */
    PeakEntry *t_e = (PeakEntry *)(resd + res_cap);             /* [cap + PYA_TABLE_PAD] (res_cap, nl_cap: 64 and 256, or what the launch needs) */
    unsigned char *tail = (unsigned char *)(t_e + cap + PYA_TABLE_PAD);
    uint16_t *nl_present = nullptr;                             /* [256]          } only with */
    float *nl_uniq = nullptr;                                   /* [PYA_MAX_UNIQ] } neutral   */
    uint8_t *resn = nullptr;                                    /* [64]           } losses    */
    if (with_nl) {
        nl_present = (uint16_t *)tail;
        nl_uniq = (float *)(nl_present + nl_cap);
        resn = (uint8_t *)(nl_uniq + PYA_MAX_UNIQ);
        tail = resn + 64;
    }
    PrefixState *pre = (PrefixState *)tail;                     /* [2][64], only if PREFIX ... */
    PrefixCompact *prc = (PrefixCompact *)tail;                 /* ... or this when `compact`  */
    NodeLds nd;                                                 /* ... or the shared-node tables (never with PREFIX) */
    nd.ntab = (const uint64_t *)tail;
    nd.run = (float *)(tail + (size_t)node_words * 8);
    nd.info = (uint16_t *)(nd.run + node_cap);
    nd.nb = nd.info + node_cap;                                 /* (node_cap is even) */
    nd.sb = nd.nb + 64;
    nd.priv = (uint32_t *)(nd.sb + 64);
    nd.b = &b;
    nd.node_cap = node_cap;
    nd.ncols = node_cols;

    if (b.status[psm] != PYA_ST_OK) return;
    const uint32_t N = b.n_sig[psm];
    if (N == 0) return;

    /* every global read the prologue needs is issued before the first LDS hand-off, so that the
     * memory round trips of the peptide, the fixed modifications and the peak table overlap */
    const Residues res = load_residues(b, cfg, psm);
    const int zmax = b.max_charge[psm];
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];
    PeakTable tab;
    stage_peak_table(b, psm, t_e, &tab);
    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = with_nl ? cfg->n_nl : 0;
    env.nl_present = nl_present;
    env.nl_uniq = nl_uniq;
    env.resd = resd;
    env.resn = resn;
    env.cnt = cnt;
    if (env.n_nl) {
        for (int i = lane; i < (int)nl_cap; i += 64) nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) nl_uniq[lane] = cfg->uniq[lane];
    }
    stage_residues(res, resd, resn);
    /* shared nodes (score_nodes_dir): general settings, one round of signatures, the shape's table behind its
     * order table */
    const int n_sites_all = __popcll(res.site_mask);
    const uint32_t ntab_words = 2u * (uint32_t)(n_sites_all + 1) * (1u + ((N + 7u) >> 3));
    const bool node_try = NODES && !PREFIX && node_cap != 0 && N <= 64 && ntab_words <= node_words && !(b.debug & 256u) &&
                          (env.n_nl != 0 || cfg->n_fwd > 1 || cfg->n_types - cfg->n_fwd > 1);
    if (node_try)
        for (uint32_t i = lane; i < ntab_words; i += 64) ((uint64_t *)nd.ntab)[i] = order[N + i];
    wave_lds_sync();
    grid_build(&tab, grid);

    env.L = res.L;
    env.zmax = zmax;
    const bool has_f = cfg->n_fwd > 0, has_b = cfg->n_fwd < cfg->n_types;
    const bool both_dirs = has_f && has_b;
    wave_lds_sync();
    /* localize looks up its few ions in the global table: leave it the grid (one 512-byte store) */
    ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)grid)[lane];

    int lut_fail = 0;
    uint32_t top_u = 0, top_n = 0, top_i = 0xffffffffu;     /* this lane's share of the summary of ws */
    const bool split = !node_try && N <= 32 && both_dirs;     /* lanes 0..31 forward, 32..63 backward */
    const bool simple = walk_is_simple(env);
    const uint32_t simple_nfrag = (uint32_t)((has_f ? 1 : 0) + (has_b ? 1 : 0)) * (uint32_t)(res.L - 1) * (uint32_t)zmax;
    const int n_sites = __popcll(res.site_mask);
    const bool shared = PREFIX && N >= 128 && n_sites >= PREFIX_SITES + 2;
    int stop[2] = {0, 0};
    if (shared) {
        /* steps [0, stop) of a direction cover exactly its first PREFIX_SITES sites */
        stop[0] = nth_set_bit(res.site_mask, PREFIX_SITES);
        stop[1] = res.L - 1 - nth_set_bit(res.site_mask, n_sites - 1 - PREFIX_SITES);
        for (int dir = 0; dir < 2; dir++) {
            if (dir == 0 ? !has_f : !has_b) continue;
            if (stop[dir] > res.L - 1) stop[dir] = res.L - 1;
            const uint64_t pbits = dir == 0 ? (uint64_t)lane : (__brevll((uint64_t)lane) >> (64 - n_sites));
            const uint64_t pmask = deposit_sites(pbits, res.site_mask);
            WalkState st = {0.f, 0u};
            uint32_t nf = 0;
            hist_clear(env);
            if (simple) walk_simple_range(env, tab, pmask, dir, true, 0, stop[dir], st);
            else walk_range(env, tab, pmask, dir, true, 0, stop[dir], st, nf);
            if (compact) {
                PrefixCompact pc;
                pc.running = st.running;
                uint64_t lo = 0ull;
#pragma unroll
                for (int d = 0; d < 8; d++) lo |= (uint64_t)hist_count(cnt, lane, d) << (d * 8);
                pc.lo = lo;
                pc.hi = hist_count(cnt, lane, 8) | (hist_count(cnt, lane, 9) << 8);
                prc[dir * 64 + lane] = pc;
            } else {
                PrefixState ps;
                ps.running = st.running;
                ps.nl_state = st.nl_state;
                ps.nfrag = nf;                              /* simple mode counts steps instead */
                ps.pad = 0;
                uint64_t hw[3] = {0ull, 0ull, 0ull};
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) hw[d >> 2] |= (uint64_t)hist_count(cnt, lane, d) << ((d & 3) * 16);
                ps.ha = hw[0];
                ps.hb = hw[1];
                ps.hc = hw[2];
                pre[dir * 64 + lane] = ps;
            }
        }
        wave_lds_sync();
    }
    for (uint32_t sbase = 0; sbase < N; sbase += 64) {
        const uint32_t s = split ? (uint32_t)(lane & 31) : sbase + lane;
        const bool active = s < N;
        const uint64_t bits = active ? order[s] : 0ull;
        const uint64_t resmask = deposit_sites(bits, res.site_mask);
        Hist ph = {0ull, 0ull, 0ull};                       /* counts inherited from the prefix table */
        uint64_t p8lo = 0ull;                               /* (compact form: 8-bit fields) */
        uint32_t p8hi = 0u;
        uint32_t nfrag = 0;
        uint32_t nacc[PYA_NTOP / 2] = {0u, 0u, 0u, 0u, 0u};   /* counts from the shared nodes (16-bit pairs, as cnt) */
        int walk_dirs = 3;                                   /* directions the walkers below still have to do */
        if (node_try) {
            walk_dirs = 0;
            for (int dir = 0; dir < 2; dir++) {
                if (dir == 0 ? !has_f : !has_b) continue;
                if (!score_nodes_dir(env, tab, nd, resmask, res.site_mask, n_sites_all, (int)N, dir, nacc, nfrag)) walk_dirs |= 1 << dir;
            }
        }
        hist_clear(env);
        if (node_try) {
            WalkState st = {0.f, 0u};
            for (int dir = 0; dir < 2; dir++) {
                if (!((walk_dirs >> dir) & 1) || (dir == 0 ? !has_f : !has_b)) continue;
                st.running = 0.f;
                st.nl_state = 0u;
                walk_range(env, tab, resmask, dir, active, 0, res.L - 1, st, nfrag);
            }
        } else if (shared) {
            WalkState sts[2] = {{0.f, 0u}, {0.f, 0u}};
            for (int dir = 0; dir < 2; dir++) {
                if (dir == 0 ? !has_f : !has_b) continue;
                const uint32_t pat = dir == 0 ? (uint32_t)(bits & 63ull)
                                              : (uint32_t)((__brevll(bits) >> (64 - n_sites)) & 63ull);
                if (compact) {
                    const PrefixCompact pc = prc[dir * 64 + pat];
                    sts[dir].running = pc.running;
                    p8lo += pc.lo;                           /* fields stay below 256: <= 63 per direction */
                    p8hi += pc.hi;
                } else {
                    const PrefixState ps = pre[dir * 64 + pat];
                    sts[dir].running = ps.running;
                    sts[dir].nl_state = ps.nl_state;
                    ph.a += ps.ha;
                    ph.b += ps.hb;
                    ph.c += ps.hc;
                    nfrag += ps.nfrag;
                }
            }
            if (simple && both_dirs && zmax == 1 && !tab.half_check) {
                /* the two directions side by side: two lookups in flight per iteration */
                walk_simple_both(env, tab, resmask, active, stop[0], res.L - 1, sts[0], stop[1], res.L - 1, sts[1]);
            } else {
                for (int dir = 0; dir < 2; dir++) {
                    if (dir == 0 ? !has_f : !has_b) continue;
                    if (simple) walk_simple_range(env, tab, resmask, dir, active, stop[dir], res.L - 1, sts[dir]);
                    else walk_range(env, tab, resmask, dir, active, stop[dir], res.L - 1, sts[dir], nfrag);
                }
            }
        } else {
            WalkState st = {0.f, 0u};
            if (split) {
                if (simple) walk_simple_range(env, tab, resmask, lane >> 5, active, 0, res.L - 1, st);
                else walk_range(env, tab, resmask, lane >> 5, active, 0, res.L - 1, st, nfrag);
            } else if (simple && both_dirs && zmax == 1 && !tab.half_check) {
                WalkState st1 = {0.f, 0u};
                walk_simple_both(env, tab, resmask, active, 0, res.L - 1, st, 0, res.L - 1, st1);
            } else {
                for (int dir = 0; dir < 2; dir++) {
                    if (dir == 0 ? !has_f : !has_b) continue;
                    st.running = 0.f;
                    st.nl_state = 0u;
                    if (simple) walk_simple_range(env, tab, resmask, dir, active, 0, res.L - 1, st);
                    else walk_range(env, tab, resmask, dir, active, 0, res.L - 1, st, nfrag);
                }
            }
        }
        if (split) nfrag += (uint32_t)__shfl_down((int)nfrag, 32, 64);   /* the backward walker sits 32 lanes up */
        if (simple) nfrag = simple_nfrag;
        wave_lds_sync();

        if (active && (!split || lane < 32)) {
            /* cumulative counts over rank (Ascore.cpp:115-118) and scores (Ascore.cpp:123-139) */
            uint32_t cum[PYA_NTOP];
            uint32_t acc = 0;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                acc += hist_count(cnt, lane, d) + (split ? hist_count(cnt, lane + 32, d) : 0u) + hist_get(ph, d) +
                       (d < 8 ? (uint32_t)(p8lo >> (d * 8)) & 0xffu : (p8hi >> ((d - 8) * 8)) & 0xffu) +
                       ((nacc[d >> 1] >> ((d & 1) * 16)) & 0xffffu);
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
            const uint32_t u = __float_as_uint(ws);              /* scores are >= 0: bit order = value order */
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
        wave_lds_sync();                                    /* columns are cleared again next round */
    }
    if (__any(lut_fail) && lane == 0) b.status[psm] = PYA_ST_LUT_RANGE;
    {
        /* largest PepScore, how many signatures share it and the first of them: localize needs no pass
         * over the scores to find its winner */
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

static inline size_t score_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact, uint32_t node_cap = 0,
                                     uint32_t node_cols = 64, uint32_t node_words = 0, uint32_t res_cap = 64, uint32_t nl_cap = 256) {
    return PYA_GRID_CELLS * 2 + PYA_NTOP / 2 * (node_cap && node_cols > 64u ? node_cols : 64u) * 4 + (size_t)res_cap * 8 +
           ((size_t)cap + PYA_TABLE_PAD) * 8 + (with_nl ? 2 * (size_t)nl_cap + PYA_MAX_UNIQ * 4 + 64 : 0) +
           (prefix ? 2 * 64 * (compact ? sizeof(PrefixCompact) : sizeof(PrefixState)) : 0) +
           (node_cap ? (size_t)node_words * 8 + (size_t)node_cap * 6 + 2 * 64 * 2 + PYA_NTOP / 2 * 64 * 4 : 0) + 64;
}

#endif
