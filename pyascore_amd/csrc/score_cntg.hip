/* score_cntg.hip -- score_signatures under the GENERAL scorer settings (neutral losses, several fragment charges, several
 * ion types per direction) from count nodes: one PSM per wavefront.
 *
 * walk_core.hip.h's count-node argument carries over when the loss variants of a fragment depend on the site assignment
 * only through the count too.  The loss state of a prefix is the multiset of the loss classes of its residues
 * (cpp/ModifiedPeptide.cpp:394-408: PowerSetSum over the losses so far); a residue that cannot be modified contributes
 * a fixed class, a modifiable one its "unmodified" or its "modified" class.  If every modifiable residue of the peptide has
 * the SAME pair of classes (phospho on S / T / Y with a loss declared for "sty": the common case), j modified residues
 * among the c modifiable ones of a prefix give j "modified" and c - j "unmodified" classes whichever they are: the node
 * (direction, step, j) has ONE loss state, hence one set of variants, and every (variant, ion type, charge) ion of the node
 * is a monotone function of the running sum -- float subtraction of the loss, double additions of the type's offsets,
 * the charge division, a narrowing -- so the envelope of the sums (cnt_envelopes) encloses each ion's m/z and one lookup
 * per ION OF A NODE decides it for every site assignment through the node (cnt_entry_f), or marks the node.
 * cfg4 (20-mer, 3 of 6 sites, b/y/c/z, charges 1-4, one loss): 2 x 4 x 19 nodes, <= 24 ions each, ~2 200 lookups where the
 * shared tree nodes of score_nodes_dir make ~6 000 and walkers 14 000; then k reads of a 24-byte entry per site
 * assignment (cumulative counts as 16-bit fields: up to 4 096 fragments per assignment).
 * A PSM whose modifiable residues differ in their classes (a terminus in the mod group, a fixed modification on a site),
 * and a site assignment through a marked node, are walked by walk_range exactly as score_signatures walks them.
 *
 * Replaces for those PSMs cpp/Ascore.cpp:53-139, cpp/ModifiedPeptide.cpp:126-150, :326-609.  Output as score_signatures.
 */
#include "score_core.hip.h"

#define CG_NODE_WORDS 4            /* LDS words per count node (cg_cumulate) */

struct CgLds {
    uint16_t *grid;
    uint32_t *cnt;       /* [PYA_NTOP / 2][64] rank histogram of the walkers (marked site assignments) */
    float2 *resd;
    PeakEntry *t_e;
    uint16_t *nl_present;
    float *nl_uniq;
    uint8_t *resn;
    uint8_t *site_pos;
    float2 *env;         /* [rows][pos_cap] */
    uint8_t *st;         /* [rows][pos_cap] loss state of the node */
    uint32_t *hist;      /* [rows][pos_cap][4]: ranks 0-9 as byte counts, word 3 = fragments | marked ions << 16 */
    uint4 *psA;          /* [rows][n_cap + 1] prefix sums at the sites' steps: cumulative counts of depths 0-7 */
    uint2 *psB;          /* ... depths 8-9, fragments | marked << 16 */
    uint4 *gA;           /* [k_cap * n_cap + 1] */
    uint2 *gB;
};
__host__ __device__ static inline size_t cg_al16(size_t v) { return (v + 15) & ~(size_t)15; }
/* r06: three work areas that take turns -- X: the nodes' rank counts (steps 3-4), then the walkers' rank histogram (step 6);
 * Y: envelopes and loss states (steps 1-3), then the per-site table G (steps 5-6); Z: the list of (node, variant) pairs (step
 * 3), then the prefix sums at the sites (steps 4-5).  Side by side they were 11.9 KB on cfg4's launch (13 wavefronts per CU);
 * taking turns 10.2 KB (15). */
__host__ __device__ static inline size_t cg_x_bytes(uint32_t pos_cap, uint32_t k_cap) {
    const size_t rows = 2 * ((size_t)k_cap + 1), h = rows * pos_cap * CG_NODE_WORDS * 4, c = PYA_NTOP / 2 * 64 * 4;
    return h > c ? h : c;
}
__host__ __device__ static inline size_t cg_y_bytes(uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap) {
    const size_t rows = 2 * ((size_t)k_cap + 1), e = rows * pos_cap * 8 + cg_al16(rows * pos_cap), g = cg_al16(((size_t)k_cap * n_cap + 1) * 24);
    return e > g ? e : g;
}
__host__ __device__ static inline size_t cg_z_bytes(uint32_t k_cap, uint32_t n_cap) { return 2 * ((size_t)k_cap + 1) * (n_cap + 1) * 24; }
__host__ __device__ static inline size_t score_cntg_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap, uint32_t nl_cap) {
    return PYA_GRID_CELLS * 2 + cg_al16(((size_t)pos_cap + 1) * 8) + ((size_t)cap + PYA_TABLE_PAD) * 8 +
           cg_al16(2 * (size_t)nl_cap) + PYA_MAX_UNIQ * 4 + 64 + 64 + cg_x_bytes(pos_cap, k_cap) + cg_y_bytes(pos_cap, k_cap, n_cap) +
           cg_z_bytes(k_cap, n_cap) + 64;
}
DEV CgLds cg_carve(unsigned char *raw, uint32_t cap, uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap, uint32_t nl_cap) {
    const size_t rows = 2 * ((size_t)k_cap + 1);
    CgLds c;
    size_t o = 0;
    c.grid = (uint16_t *)raw;
    o += PYA_GRID_CELLS * 2;
    c.resd = (float2 *)(raw + o);
    o += cg_al16(((size_t)pos_cap + 1) * 8);
    c.t_e = (PeakEntry *)(raw + o);
    o += ((size_t)cap + PYA_TABLE_PAD) * 8;
    c.nl_present = (uint16_t *)(raw + o);
    o += cg_al16(2 * (size_t)nl_cap);
    c.nl_uniq = (float *)(raw + o);
    o += PYA_MAX_UNIQ * 4;
    c.resn = raw + o;
    o += 64;
    c.site_pos = raw + o;
    o += 64;
    c.env = (float2 *)(raw + o);                             /* Y */
    c.st = raw + o + rows * pos_cap * 8;
    c.gA = (uint4 *)(raw + o);
    c.gB = (uint2 *)(raw + o + ((size_t)k_cap * n_cap + 1) * 16);
    o += cg_y_bytes(pos_cap, k_cap, n_cap);
    c.hist = (uint32_t *)(raw + o);                          /* X */
    c.cnt = (uint32_t *)(raw + o);
    o += cg_x_bytes(pos_cap, k_cap);
    c.psA = (uint4 *)(raw + o);                              /* Z */
    o += rows * (n_cap + 1) * 16;
    c.psB = (uint2 *)(raw + o);
    return c;
}

/* the node's rank counts (BYTE fields, ranks 0-9 in words 0-2; word 3 = fragments | marked ions << 16) -> cumulative counts
 * over the ranks as 16-bit fields in words 0-4, word 5 = word 3.  (r06: 16 bytes per node instead of 24 -- a node has at most
 * ion types x charges x loss variants ions, and a launch whose bound exceeds 255 walks instead: the kernel's LDS decides how
 * many wavefronts a CU holds, and its time follows that.) */
DEV void cg_cumulate(const uint32_t *h, uint32_t out[6]) {
    uint32_t acc = 0;
#pragma unroll
    for (int w = 0; w < 5; w++) {
        const int r0 = 2 * w, r1 = 2 * w + 1;
        const uint32_t lo = acc + ((h[r0 >> 2] >> ((r0 & 3) * 8)) & 0xffu), hi = lo + ((h[r1 >> 2] >> ((r1 & 3) * 8)) & 0xffu);
        out[w] = lo | (hi << 16);
        acc = hi;
    }
    out[5] = h[3];
}

DEV void score_cntg_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t cap, uint32_t pos_cap, uint32_t k_cap,
                         uint32_t n_cap, uint32_t nl_cap) {
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const CgLds c = cg_carve(lds_raw, cap, pos_cap, k_cap, n_cap, nl_cap);
    /* (r06: the prologue's loads in two rounds -- device_common.hip.h: load_desc) */
    const LetterRegs letters = load_letter_regs(cfg);
    const PsmDesc dsc = load_desc(b, psm);
    const int status0 = b.status[psm];
    const int R0 = (int)b.ret_n[psm];
    if (status0 != PYA_ST_OK) return;
    const uint32_t N = dsc.N;
    if (N == 0) return;
    STAMP_BEGIN();
    const Residues res = load_residues_desc(b, cfg, dsc, letters);
    const int zmax = dsc.zmax;
    const uint64_t *order = b.order_tab + dsc.order_off;
    const int64_t s0 = dsc.sig0;
    const int L = res.L, Lm1 = L - 1, k = dsc.k, n_sites = __popcll(res.site_mask);
    PeakTable tab;
    stage_peak_table_at(b, dsc.ret0, R0, c.t_e, &tab);
    WalkEnv env;
    env.cfg = cfg;
    env.n_nl = cfg->n_nl;
    env.nl_present = c.nl_present;
    env.nl_uniq = c.nl_uniq;
    env.resd = c.resd;
    env.resn = c.resn;
    env.cnt = c.cnt;
    env.L = L;
    env.zmax = zmax;
    if (env.n_nl) {
        for (int i = lane; i < (int)nl_cap; i += 64) c.nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) c.nl_uniq[lane] = cfg->uniq[lane];
    }
    stage_residues(res, c.resd, c.resn);
    const bool is_site = (res.site_mask >> lane) & 1ull;
    if (is_site) c.site_pos[mask_rank(res.site_mask)] = (uint8_t)lane;
    wave_lds_sync();
    grid_build(&tab, c.grid);
    wave_lds_sync();
    ((uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane] = ((const uint64_t *)c.grid)[lane];
    STAMP_T(b, 25, );

    const int n_f = cfg->n_fwd, n_b = cfg->n_types - cfg->n_fwd;
    const int t_max = n_f > n_b ? n_f : n_b;
    const uint64_t types64 = load_types64(cfg);
    const int rows = 2 * (k + 1);
    /* every modifiable residue the same pair of loss classes?  (else: every site assignment is walked) */
    const uint32_t first_nl = (uint32_t)__builtin_amdgcn_readlane((int)res.nl, res.site_mask ? __builtin_ctzll(res.site_mask) : 0);
    const bool uniform = !__any(is_site && res.nl != first_nl) && !(b.debug & 0x40000000u) && k + 1 <= 31 && n_sites <= 32 &&
                         (uint32_t)k <= k_cap && (uint32_t)n_sites <= n_cap &&
                         (uint32_t)t_max * (uint32_t)zmax * (env.n_nl ? (uint32_t)PYA_MAX_UNIQ : 1u) <= 255u;   /* (byte counts per node) */
    bool tables = false;                                     /* the count tables were built: site assignments read them */
    if (uniform) {
        /* 1. envelopes of the running sums; 2. the loss state of every node, from the chain with the first j sites modified */
        cnt_envelopes(res, k, pos_cap, c.env);
        if (env.n_nl) {
            const int d = lane >> 5, j = lane & 31;
            if (j <= k) {
                const uint64_t first_j = (1ull << j) - 1ull;
                const uint64_t pbits = d == 0 ? first_j : (first_j << (n_sites - j));
                const uint64_t resmask = deposit_sites(pbits, res.site_mask);
                uint32_t st = 0;
                uint8_t *out = c.st + (size_t)(d * (k + 1) + j) * pos_cap;
                for (int step = 0; step < Lm1; step++) {
                    const int ri = d ? L - 1 - step : step;
                    const bool mod = (resmask >> ri) & 1ull;
                    const uint32_t nlp = c.resn[ri];
                    const uint32_t cls = mod ? (nlp >> 4) : (nlp & 15u);
                    if (cls) st = nl_bump(st, cls);
                    out[step] = (uint8_t)st;
                }
            }
        }
        for (int i = lane; i < rows * (int)pos_cap * CG_NODE_WORDS; i += 64) c.hist[i] = 0u;
        wave_lds_sync();
        STAMP_T(b, 26, );
        /* 3. one lookup per (node, variant, ion type, charge).  The (reachable node, loss variant) pairs are listed first: a
         * third of the (row, step) pairs have fewer modifiable residues than j, and a node has one to three variants (more
         * with more loss masses) -- a lane per node with its variants in a loop ran every wavefront to its longest node.  A lane
         * takes a (node, variant, ion type) item, the charges in a wave-uniform inner loop (so that the charge division runs
         * for the charges that need it only, not for every lane's own). */
        uint16_t *nlist = (uint16_t *)c.psA;                     /* (the prefix sums' and G's room: not written before step 4) */
        const int n_room = (int)(cg_z_bytes(k_cap, n_cap) / 2u);
        int n_ent = 0;
        for (int row = 0; row < rows; row++) {
            uint32_t pm = 0u;
            if (lane < Lm1) {
                const size_t node = (size_t)row * pos_cap + lane;
                const float2 lh = c.env[node];
                if (lh.x <= lh.y) pm = env.n_nl ? (uint32_t)c.nl_present[c.st[node]] : 1u;
            }
            while (__any(pm != 0u)) {
                const bool has = pm != 0u;
                const int v = has ? __builtin_ctz(pm) : 0;
                pm &= pm - 1u;
                const uint64_t hm = __ballot(has);
                const int at = n_ent + mask_rank(hm);
                if (has && at < n_room) nlist[at] = (uint16_t)((row * 64 + lane) | (v << 12));
                n_ent += __popcll(hm);
            }
        }
        tables = n_ent <= n_room;                                /* (else: every site assignment is walked) */
        wave_lds_sync();
        if (tables) {
            const FastDiv divT = fastdiv_make((uint32_t)t_max);
            const uint32_t items = (uint32_t)n_ent * (uint32_t)t_max;
            for (uint32_t base = 0; base < items; base += 64) {
                const uint32_t i = base + (uint32_t)lane;
                const uint32_t ni = fastdiv(i, divT);
                const int t = (int)(i - ni * (uint32_t)t_max);
                const uint32_t code = i < items ? (uint32_t)nlist[ni] : 0u;
                const uint32_t row = (code >> 6) & 63u, s = code & 63u, v = code >> 12;
                const int d = row >= (uint32_t)(k + 1) ? 1 : 0;
                const int my_types = d ? n_b : n_f;
                const bool on = i < items && t < my_types;
                const size_t node = (size_t)row * pos_cap + s;
                const float2 lh = c.env[node];
                double A, B;
                type_constants(type_at(types64, (d ? n_f : 0) + (t < my_types ? t : 0)), &A, &B);
                uint32_t *hn = c.hist + node * CG_NODE_WORDS;
                if (on && t == 0) atomicAdd(&hn[3], (uint32_t)(my_types * zmax));
                const float loss = env.n_nl ? c.nl_uniq[v] : 0.f;
                const float x_lo = env.n_nl ? lh.x - loss : lh.x, x_hi = env.n_nl ? lh.y - loss : lh.y;   /* float subtract (:572), monotone */
                const double m_lo = ((double)x_lo + A) - B, m_hi = ((double)x_hi + A) - B;
                /* (the items follow the rows, and the rows j = 0 and j = k have one chain per node: whole rounds of point
                 * envelopes, which need one m/z and no band) */
                if (__all(!on || lh.x == lh.y)) {
                    for (int z = 1; z <= zmax; z++) {
                        const uint32_t rk = cnt_entry_1(tab, charge_mz(m_lo, z));
                        if (on && rk < (uint32_t)PYA_NTOP) atomicAdd(&hn[rk >> 2], 1u << ((rk & 3u) * 8u));
                    }
                } else {
                    for (int z = 1; z <= zmax; z++) {
                        const uint32_t ent = cnt_entry_f(tab, charge_mz(m_lo, z), charge_mz(m_hi, z));
                        const uint32_t rk = ent & 15u;
                        if (on && rk < (uint32_t)PYA_NTOP) atomicAdd(&hn[rk >> 2], 1u << ((rk & 3u) * 8u));
                        if (on && (ent & CNT_MARK)) atomicAdd(&hn[3], 1u << 16);
#ifdef PYA_STAMPS                                              /* diagnostic build: ions of nodes with an interval envelope looked up / marked (56, 57) */
                        if (b.stamps && on) {
                            atomicAdd(&b.stamps[56], 1ull);
                            if (ent & CNT_MARK) atomicAdd(&b.stamps[57], 1ull);
                        }
#endif
                    }
                }
            }
        }
        wave_lds_sync();
        STAMP_T(b, 27, );
        if (tables) {
        /* 4. prefix sums over the steps of every row, taken at the sites' steps (score_cnt.hip: site_prefix_sums): two rows
         * per scan, one per half of the wavefront, for peptides of up to 33 residues */
        {
            const bool two = Lm1 <= 32 && n_sites <= 32;
            const int half = two ? lane >> 5 : 0, sl = two ? lane & 31 : lane, hbase = two ? (lane & 32) : 0;
            const int pos = sl < n_sites ? (int)c.site_pos[sl] : 0;
            for (int r0 = 0; r0 < rows; r0 += two ? 2 : 1) {
                const int row = r0 + half;
                const bool row_on = row < rows;
                const int d = row >= k + 1 ? 1 : 0;
                uint32_t w[6] = {0u, 0u, 0u, 0u, 0u, 0u};
                if (sl < Lm1 && row_on) cg_cumulate(c.hist + ((size_t)row * pos_cap + sl) * CG_NODE_WORDS, w);
#pragma unroll
                for (int x = 0; x < 6; x++) w[x] = two ? wave_incl_scan_u32<true>(w[x]) : wave_incl_scan_u32<false>(w[x]);
                const int st = d ? Lm1 - pos : pos;
                const int e = st < Lm1 ? st : Lm1;
                const int src = hbase + (e > 0 ? e - 1 : 0), last = hbase + (Lm1 > 0 ? Lm1 - 1 : 0);
                uint32_t pv[6], tv[6];
#pragma unroll
                for (int x = 0; x < 6; x++) {
                    pv[x] = (uint32_t)__shfl((int)w[x], src, 64);
                    if (e == 0) pv[x] = 0u;
                    tv[x] = (uint32_t)__shfl((int)w[x], last, 64);
                }
                if (row_on) {
                    const size_t o2 = (size_t)row * (n_sites + 1);
                    if (sl < n_sites) {
                        c.psA[o2 + sl] = make_uint4(pv[0], pv[1], pv[2], pv[3]);
                        c.psB[o2 + sl] = make_uint2(pv[4], pv[5]);
                    }
                    if (sl == 0) {
                        c.psA[o2 + n_sites] = make_uint4(tv[0], tv[1], tv[2], tv[3]);
                        c.psB[o2 + n_sites] = make_uint2(tv[4], tv[5]);
                    }
                }
            }
        }
        wave_lds_sync();
        STAMP_T(b, 28, );
        /* 5. G(t, site) and the constant (walk_core.hip.h) */
        {
            const int W = n_sites + 1;
            for (int i = lane; i <= k * n_sites; i += 64) {
                uint4 ga;
                uint2 gb;
                if (i == k * n_sites) {
                    const uint4 a = c.psA[(size_t)k * W + n_sites], q = c.psA[(size_t)(k + 1 + k) * W + n_sites];
                    const uint2 a2 = c.psB[(size_t)k * W + n_sites], q2 = c.psB[(size_t)(k + 1 + k) * W + n_sites];
                    ga = make_uint4(a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w);
                    gb = make_uint2(a2.x + q2.x, a2.y + q2.y);
                } else {
                    const int t = i / n_sites + 1, site = i - (t - 1) * n_sites, tb = k + 1 - t;
                    const size_t f0 = (size_t)(t - 1) * W + site, f1 = (size_t)t * W + site, b0 = (size_t)(k + 1 + tb - 1) * W + site, b1 = (size_t)(k + 1 + tb) * W + site;
                    const uint4 A0 = c.psA[f0], A1 = c.psA[f1], B0 = c.psA[b0], B1 = c.psA[b1];
                    const uint2 a0 = c.psB[f0], a1 = c.psB[f1], bb0 = c.psB[b0], bb1 = c.psB[b1];
                    ga = make_uint4((A0.x - A1.x) + (B0.x - B1.x), (A0.y - A1.y) + (B0.y - B1.y), (A0.z - A1.z) + (B0.z - B1.z), (A0.w - A1.w) + (B0.w - B1.w));
                    gb = make_uint2((a0.x - a1.x) + (bb0.x - bb1.x), (a0.y - a1.y) + (bb0.y - bb1.y));
                }
                c.gA[i] = ga;
                c.gB[i] = gb;
            }
        }
        wave_lds_sync();
        }
    }

    STAMP_T(b, 29, );
    /* 6. the site assignments */
    const bool has_f = n_f > 0, has_b = n_b > 0;
    int lut_fail = 0;
    uint32_t top_u = 0, top_n = 0, top_i = 0xffffffffu;
    for (uint32_t sbase = 0; sbase < N; sbase += 64) {
        const uint32_t s = sbase + (uint32_t)lane;
        const bool active = s < N;
        const uint64_t bits = active ? order[s] : 0ull;
        uint32_t w[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        bool walk = active;                                  /* (no tables: everybody walks) */
        if (tables) {
            const uint4 ca = c.gA[k * n_sites];
            const uint2 cb = c.gB[k * n_sites];
            w[0] = ca.x; w[1] = ca.y; w[2] = ca.z; w[3] = ca.w; w[4] = cb.x; w[5] = cb.y;
            uint32_t m = (uint32_t)bits;
            for (int t = 0; t < k; t++) {
                const int site = active ? __builtin_ctz(m) : 0;
                m &= m - 1u;
                const uint4 ga = c.gA[t * n_sites + site];
                const uint2 gb = c.gB[t * n_sites + site];
                w[0] += ga.x; w[1] += ga.y; w[2] += ga.z; w[3] += ga.w; w[4] += gb.x; w[5] += gb.y;
            }
            walk = active && (w[5] >> 16) != 0u;
        }
#ifdef PYA_STAMPS                                              /* ... site assignments / walked ones (58, 59) */
        if (b.stamps && active) {
            atomicAdd(&b.stamps[58], 1ull);
            if (walk) atomicAdd(&b.stamps[59], 1ull);
        }
#endif
        uint32_t cum[PYA_NTOP];
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d++) cum[d] = (w[d >> 1] >> ((d & 1) * 16)) & 0xffffu;
        uint32_t nfrag = w[5] & 0xffffu;
        if (__any(walk)) {
            /* a marked node on its path, or a peptide whose sites differ in their loss classes: walked with its own sums */
            const uint64_t resmask = deposit_sites(bits, res.site_mask);
            hist_clear(env);
            uint32_t nf = 0;
            for (int dir = 0; dir < 2; dir++) {
                if (dir == 0 ? !has_f : !has_b) continue;
                WalkState st = {0.f, 0u};
                walk_range(env, tab, resmask, dir, walk, 0, L - 1, st, nf);
            }
            wave_lds_sync();
            if (walk) {
                uint32_t acc = 0;
#pragma unroll
                for (int d = 0; d < PYA_NTOP; d++) {
                    acc += hist_count(c.cnt, lane, d);
                    cum[d] = acc;
                }
                nfrag = nf;
            }
            wave_lds_sync();
        }
        if (active) {
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

#ifndef SCORE_CNTG_WAVES
#define SCORE_CNTG_WAVES 5
#endif
__global__ __launch_bounds__(64, SCORE_CNTG_WAVES) void pya_score_cntg_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap,
                                                                            uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap, uint32_t nl_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    score_cntg_body(b, psm_ids[xcd_slot(blockIdx.x, n_ids)], lds_raw, cap, pos_cap, k_cap, n_cap, nl_cap);
}

extern "C" size_t pya_score_cntg_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap, uint32_t nl_cap) {
    return score_cntg_lds_bytes(cap, pos_cap, k_cap, n_cap, nl_cap);
}

extern "C" int pya_launch_score_cntg(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap, uint32_t k_cap,
                                     uint32_t n_cap, uint32_t nl_cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = score_cntg_lds_bytes(cap, pos_cap, k_cap, n_cap, nl_cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_score_cntg_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_score_cntg_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, pos_cap, k_cap, n_cap, nl_cap);
    return (int)hipGetLastError();
}
