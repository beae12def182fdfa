/* walk_core.hip.h -- the per-(signature, direction) fragment walker shared by
 * score_signatures.hip and fused_small.hip.  See score_signatures.hip for the notes. */
#ifndef PYA_WALK_CORE_H
#define PYA_WALK_CORE_H
#include "device_common.hip.h"

struct WalkEnv {
    const DevConfig *cfg;
    const uint16_t *nl_present;
    const float *nl_uniq;
    int n_nl, L, zmax;
};

/* Walks one direction per lane (`dir` may differ between lanes: the residue of a step is read
 * for both directions with two v_readlane and selected).  Adds to h / nfrag. */
/* resumable state of a walker: float32 running sum and neutral-loss stack state */
struct WalkState {
    float running;
    uint32_t nl_state;
};

DEV void walk_range(const WalkEnv &e, const Residues &res, const PeakTable &tab, uint64_t resmask, int dir,
                    bool active, int step_begin, int step_end, WalkState &st, Hist &h, uint32_t &nfrag) {
    const DevConfig *cfg = e.cfg;
    const int n_f = cfg->n_fwd, n_b = cfg->n_types - cfg->n_fwd;
    const int my_types = dir == 0 ? n_f : n_b;
    const int t_base = dir == 0 ? 0 : n_f;
    const bool any_f = __any(active && dir == 0), any_b = __any(active && dir == 1);
    const int t_max = (any_f && any_b) ? (n_f > n_b ? n_f : n_b) : (any_f ? n_f : n_b);
    const uint64_t types64 = load_types64(cfg);
    float running = st.running;                            /* 0 at the start: 0 + r == r exactly */
    uint32_t nl_state = st.nl_state;
    for (int step = step_begin; step < step_end; step++) {
        const int i_f = step, i_b = e.L - 1 - step;                      /* wave-uniform */
        float m0 = 0.f, m1 = 0.f;
        uint32_t nlp = 0;
        if (any_f) {
            m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), i_f));
            m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), i_f));
            nlp = (uint32_t)__builtin_amdgcn_readlane((int)res.nl, i_f);
        }
        if (any_b) {
            const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), i_b));
            const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), i_b));
            const uint32_t bn = (uint32_t)__builtin_amdgcn_readlane((int)res.nl, i_b);
            m0 = dir ? b0 : m0;
            m1 = dir ? b1 : m1;
            nlp = dir ? bn : nlp;
        }
        const int i = dir ? i_b : i_f;
        const bool mod = (resmask >> i) & 1ull;
        const float r = mod ? m1 : m0;
        running = r + running;                                           /* ModifiedPeptide.cpp:385-389 */
        uint32_t pm = active ? 1u : 0u;
        if (e.n_nl) {
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
                        hist_add(h, match_rank_lds(tab, f));
                        nfrag++;
                    }
                }
            }
        }
    }
    st.running = running;
    st.nl_state = nl_state;
}

DEV void walk(const WalkEnv &e, const Residues &res, const PeakTable &tab, uint64_t resmask, int dir,
              bool active, Hist &h, uint32_t &nfrag) {
    WalkState st = {0.f, 0u};
    walk_range(e, res, tab, resmask, dir, active, 0, e.L - 1, st, h, nfrag);
}

/* Fast path of `walk` for the common scorer settings -- no neutral losses, charge 1, at most
 * one ion type per direction (BASELINE cfg1/2/3/5): straight-line code per residue step.
 * `tmask` bit t = "the t-th residue in THIS lane's travel direction is modified". */
DEV void walk_simple_range(const WalkEnv &e, const Residues &res, const PeakTable &tab, uint64_t resmask,
                           int dir, bool active, int step_begin, int step_end, WalkState &st, Hist &h,
                           uint32_t &nfrag) {
    const DevConfig *cfg = e.cfg;
    const int L = e.L;
    /* per-lane ion-type constants (the type letters are wave-uniform scalars) */
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (cfg->n_fwd > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (cfg->n_fwd < cfg->n_types) type_constants(cfg->types[cfg->n_fwd], &Ab, &Bb);
    const double A = dir ? Ab : Af, B = dir ? Bb : Bf;
    const uint64_t tmask = dir ? (__brevll(resmask) >> (64 - L)) : resmask;
    const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
    float running = st.running;                            /* 0 at the start: 0 + r == r exactly */
    for (int step = step_begin; step < step_end; step++) {
        const int i_b = L - 1 - step;
        const float f0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), step));
        const float f1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), step));
        const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), i_b));
        const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), i_b));
        const uint32_t word = step < 32 ? tlo : thi;       /* wave-uniform choice */
        const bool mod = (word >> (step & 31)) & 1u;
        const float m0 = dir ? b0 : f0, m1 = dir ? b1 : f1;
        const float r = mod ? m1 : m0;
        running = r + running;                             /* ModifiedPeptide.cpp:385-389 */
        const double m = ((double)running + A) - B;
        const float f = (float)(m + 1.007825);
        const int rk = match_rank_lds(tab, f);
        if (active) hist_add(h, rk);
    }
    if (active && step_end > step_begin) nfrag += (uint32_t)(step_end - step_begin);
    st.running = running;
}

DEV void walk_simple(const WalkEnv &e, const Residues &res, const PeakTable &tab, uint64_t resmask, int dir,
                     bool active, Hist &h, uint32_t &nfrag) {
    WalkState st = {0.f, 0u};
    walk_simple_range(e, res, tab, resmask, dir, active, 0, e.L - 1, st, h, nfrag);
}

DEV bool walk_is_simple(const WalkEnv &e) {
    const int n_f = e.cfg->n_fwd, n_b = e.cfg->n_types - e.cfg->n_fwd;
    return e.n_nl == 0 && e.zmax == 1 && n_f <= 1 && n_b <= 1;
}

/* walker of the opposite direction sits 32 lanes up: fold it into lanes 0..31 */
DEV void fold_upper_half(Hist &h, uint32_t &nfrag) {
    h.a += __shfl_down(h.a, 32, 64);
    h.b += __shfl_down(h.b, 32, 64);
    h.c += __shfl_down(h.c, 32, 64);
    nfrag += (uint32_t)__shfl_down((int)nfrag, 32, 64);
}


#endif
