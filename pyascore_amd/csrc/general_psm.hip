/* general_psm.hip -- the whole Ascore path of ONE PSM by one wavefront, without the size limits the fast kernels
 * buy their speed with: peptides up to 255 residues (a 64-bit residue mask and one residue per lane are the data
 * layout of every other kernel), any number of site assignments the workspace holds (the others keep their sort
 * room and pre-sort tables in LDS), fragment lists of up to 8 192 ions per ion type.
 *
 * It is the reference's algorithm in the reference's own shape, laid out for a wavefront only where that is free:
 *   residues, fixed modifications, neutral-loss classes        cpp/ModifiedPeptide.cpp:24-79
 *   counts per site assignment (one assignment per lane)       cpp/Ascore.cpp:53-121, cpp/ModifiedPeptide.cpp:326-609
 *   PepScores from the host-built score table                  cpp/Ascore.cpp:123-139, cpp/Util.cpp:16-83
 *   the front of std::sort (wave emulation in global memory)   cpp/Ascore.cpp:141-146
 *   single-move competitors, the best of every modified site   cpp/Ascore.cpp:212-254
 *   site-determining ions: both lists, sorted, greedy walk     cpp/ModifiedPeptide.cpp:259-320
 *   Ascores                                                    cpp/Ascore.cpp:157-210
 * n_top (peaks retained per window = depths scored) is a run-time value here, 10 to 16: such depths are counted and
 * scored, the first ten weighted, all of them searched for the depth of an Ascore (Ascore.cpp:15-36, :123-139, :164-172);
 * a scorer created with n_top > 10 sends every PSM here.
 * Binning is not repeated: the PSM's spectrum goes through bin_spectra like every other one.  The host sends a PSM
 * here only when it exceeds a limit of the fast kernels (host_plan.cpp: plan_create_impl); a batch of ordinary PSMs never
 * launches this kernel.  Speed is not a goal: the lookups are binary searches in the workspace table, the lists are
 * ranked by counting, the walk is one lane's.
 */
#include "localize_core.hip.h"

#define GEN_NO_MATCH 255
#define GEN_MAX_SITES 64

/* min rank over retained peaks p with f32(f - err) < p < f32(f + err) and f >= p - 0.5 (ModifiedPeptide.cpp:126-150) */
DEV int gen_match_rank(const PeakEntry *e, int n, float f, float err, bool half_check) {
    const float lo = f - err, hi = f + err;
    int a = 0, b = n;
    while (a < b) {                                           /* first entry above lo */
        const int m = (a + b) >> 1;
        if (e[m].mz > lo) b = m;
        else a = m + 1;
    }
    int best = GEN_NO_MATCH;
    for (int i = a; i < n; i++) {
        const PeakEntry x = e[i];
        if (!(x.mz < hi)) break;
        if (!half_check || (double)f >= (double)x.mz - 0.5) best = (int)x.rank < best ? (int)x.rank : best;
    }
    return best;
}

/* LDS of one wavefront (l_cap residues, list_cap ions per list) */
struct GenLds {
    float *m0, *m1;              /* [l_cap] */
    float *run;                  /* [2][l_cap] running sums of the two signatures being compared, current direction */
    uint32_t *cpre;              /* [2][l_cap] ions (x charges) before the prefix */
    uint64_t *pm;                /* [2][l_cap] loss sums present (bit v = uniq[v]) */
    uint8_t *nl0, *nl1, *sor;    /* [l_cap] loss class unmodified / modified, site index of the residue (255: none) */
    uint8_t *site_pos;           /* [64] residue of the j-th modifiable one */
    float *uniq;                 /* [PYA_MAX_UNIQ_WIDE] the distinct sums of <= 2 neutral losses, [0] = none */
    uint8_t *cand;               /* [3][PYA_MAX_NL_CANDS] classes a, b (255: a alone) and sum number of every candidate */
    uint32_t *site_max, *site_tie;   /* [64] */
    unsigned long long *site_alt;    /* [64] */
    float *site_asc;             /* [64] */
    uint32_t *misc;              /* [16] counters */
    float *sc;                   /* [2][PYA_NTOP_MAX] depth scores of the two signatures */
    float *la, *lb, *sa, *sb;    /* [list_cap] each: the two lists, unsorted and sorted */
    uint8_t *ha, *hb;            /* [list_cap] the sorted ion matched a peak of rank <= depth */
};
__host__ __device__ static inline size_t gen_lds_bytes(uint32_t l_cap, uint32_t list_cap) {
    const size_t lc = (l_cap + 3u) & ~3u;
    return lc * (4 + 4 + 2 * 4 + 2 * 4 + 2 * 8 + 3) + 64 + 3 * PYA_MAX_NL_CANDS + 4 + PYA_MAX_UNIQ_WIDE * 4 + 64 * (4 + 4 + 8 + 4) + 64 +
           2 * PYA_NTOP_MAX * 4 + 8 + (size_t)list_cap * (4 * 4 + 2) + 64;
}
DEV GenLds gen_carve(unsigned char *raw, uint32_t l_cap, uint32_t list_cap) {
    const size_t lc = (l_cap + 3u) & ~3u;
    GenLds g;
    g.site_alt = (unsigned long long *)raw;
    g.pm = (uint64_t *)(g.site_alt + 64);
    g.m0 = (float *)(g.pm + 2 * lc);
    g.m1 = g.m0 + lc;
    g.run = g.m1 + lc;
    g.cpre = (uint32_t *)(g.run + 2 * lc);
    g.uniq = (float *)(g.cpre + 2 * lc);
    g.site_max = (uint32_t *)(g.uniq + PYA_MAX_UNIQ_WIDE);
    g.site_tie = g.site_max + 64;
    g.site_asc = (float *)(g.site_tie + 64);
    g.misc = (uint32_t *)(g.site_asc + 64);
    g.sc = (float *)(g.misc + 16);
    g.la = g.sc + 2 * PYA_NTOP_MAX + 2;
    g.lb = g.la + list_cap;
    g.sa = g.lb + list_cap;
    g.sb = g.sa + list_cap;
    g.nl0 = (uint8_t *)(g.sb + list_cap);
    g.nl1 = g.nl0 + lc;
    g.sor = g.nl1 + lc;
    g.site_pos = g.sor + lc;
    g.cand = g.site_pos + 64;
    g.ha = g.cand + 3 * PYA_MAX_NL_CANDS + 4;
    g.hb = g.ha + list_cap;
    return g;
}

DEV void gen_sync() {           /* lanes hand data over through LDS and through the workspace */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

/* is residue ri modified in the signature `bits` (bit j = j-th modifiable residue)? */
DEV bool gen_modified(const GenLds &g, uint64_t bits, int ri) {
    const uint32_t j = g.sor[ri];
    return j != 255u && ((bits >> j) & 1ull);
}

/* which loss sums exist for a fragment whose residues carried the loss classes counted in `st` (2 bits per class,
 * saturating at 2): none, every class seen, every pair of classes seen, a class twice when it was seen twice --
 * PowerSetSum(stack, 2), cpp/Util.cpp:95-141, as the set of distinct values (host_tables.cpp: build_dev_config) */
DEV uint64_t gen_present(const GenLds &g, int n_cand, uint32_t st) {
    uint64_t m = 1ull;
    for (int i = 0; i < n_cand; i++) {
        const uint32_t a = g.cand[i], b2 = g.cand[PYA_MAX_NL_CANDS + i];
        const uint32_t ca = (st >> (2u * a)) & 3u;
        const bool ok = b2 == 255u ? ca >= 1u : (b2 == a ? ca >= 2u : (ca >= 1u && ((st >> (2u * b2)) & 3u) >= 1u));
        if (ok) m |= 1ull << g.cand[2 * PYA_MAX_NL_CANDS + i];
    }
    return m;
}

/* Running sums, loss sums present and ion offsets of one signature along one direction (one lane):
 * ModifiedPeptide.cpp:385-408.  Returns the number of (prefix, loss variant) pairs. */
DEV uint32_t gen_prefix_table(const GenLds &g, const DevConfig *cfg, uint64_t bits, int L, int dir, int slot, uint32_t lc) {
    float running = 0.f;
    uint32_t st = 0, cnt = 0;
    uint64_t pm = 1ull;
    for (int step = 0; step + 1 < L; step++) {
        const int ri = dir ? L - 1 - step : step;
        const bool mod = gen_modified(g, bits, ri);
        const float r = mod ? g.m1[ri] : g.m0[ri];
        running = r + running;
        if (cfg->n_nl) {
            const uint32_t cls = mod ? g.nl1[ri] : g.nl0[ri];
            if (cls) {
                const uint32_t st2 = nl_bump(st, cls);
                if (st2 != st) pm = gen_present(g, cfg->n_cand, st2);
                st = st2;
            }
        }
        g.run[slot * lc + step] = running;
        g.pm[slot * lc + step] = pm;
        g.cpre[slot * lc + step] = cnt;
        cnt += (uint32_t)__popcll(pm);
    }
    return cnt;
}

/* the ions of one (signature slot, ion type), every charge and loss variant, into `out`: one prefix per lane and trip */
DEV void gen_fill_list(const GenLds &g, const DevConfig *cfg, int L, int zmax, int slot, uint32_t lc, double A, double B, float *out) {
    for (int step = lane_id(); step + 1 < L; step += 64) {
        const float running = g.run[slot * lc + step];
        uint64_t pm = g.pm[slot * lc + step];
        uint32_t at = g.cpre[slot * lc + step] * (uint32_t)zmax;
        while (pm) {
            const int v = __builtin_ctzll(pm);
            pm &= pm - 1;
            const float x = running - (cfg->n_nl ? g.uniq[v] : 0.f);
            const double m = ((double)x + A) - B;
            for (int z = 1; z <= zmax; z++) out[at++] = charge_mz(m, z);
        }
    }
}

/* ascending order by counting: position = ions below + equal ions before (any correct sort leaves the same values) */
DEV void gen_rank_sort(const float *in, float *out, int n) {
    for (int i = lane_id(); i < n; i += 64) {
        const float x = in[i];
        int pos = 0;
        for (int j = 0; j < n; j++) {
            const float y = in[j];
            pos += (y < x || (y == x && j < i)) ? 1 : 0;
        }
        out[pos] = x;
    }
}

/* residues, fixed modifications, neutral-loss classes, the list of modifiable residues (ModifiedPeptide.cpp:24-79);
 * zeroes the per-site work areas.  Returns the number of modifiable residues. */
DEV int gen_setup_residues(const BatchDev &b, const DevConfig *cfg, const GenLds &g, uint32_t psm, int64_t pep0, int L) {
    const int lane = lane_id();
    for (int i = lane; i < L; i += 64) {
        const uint32_t li = ((uint32_t)b.pep[pep0 + i] - 'A') & 31u;
        const float m0 = cfg->res_mass[li];
        const bool modifiable = cfg->res_modifiable[li] || (cfg->allow_n && i == 0) || (cfg->allow_c && i == L - 1);
        g.m0[i] = m0;
        g.m1[i] = m0 + cfg->mod_mass;
        g.nl0[i] = cfg->nl_upper[li];
        g.nl1[i] = modifiable ? cfg->nl_lower[li] : 0;
        g.sor[i] = modifiable ? 0 : 255;
    }
    if (lane < PYA_MAX_UNIQ_WIDE) g.uniq[lane] = cfg->uniq_w[lane];
    if (lane < PYA_MAX_NL_CANDS) {
        g.cand[lane] = cfg->cand_a[lane];
        g.cand[PYA_MAX_NL_CANDS + lane] = cfg->cand_b[lane];
        g.cand[2 * PYA_MAX_NL_CANDS + lane] = cfg->cand_u[lane];
    }
    if (lane < GEN_MAX_SITES) {
        g.site_max[lane] = 0;
        g.site_tie[lane] = 0;
        g.site_alt[lane] = 0ull;
        g.site_asc[lane] = __builtin_huge_valf();
    }
    if (lane < 16) g.misc[lane] = 0;
    gen_sync();
    if (lane == 0) {
        for (int64_t a = b.aux_off[psm]; a < b.aux_off[psm + 1]; a++) {       /* fixed modifications, in their order */
            const uint32_t pos = b.aux_pos[a];
            const float am = b.aux_mass[a];
            const int idx = pos > 0 ? (int)pos - 1 : 0;
            if (idx < L) {
                const uint32_t li = ((uint32_t)b.pep[pep0 + idx] - 'A') & 31u;
                g.m0[idx] += am;
                g.m1[idx] += am;
                if (cfg->nl_lower[li]) g.nl0[idx] = cfg->nl_lower[li];
            }
        }
        int j = 0;
        for (int i = 0; i < L; i++)
            if (g.sor[i] != 255) {
                if (j < GEN_MAX_SITES) g.site_pos[j] = (uint8_t)i;
                g.sor[i] = (uint8_t)j;
                j++;
            }
        g.misc[0] = (uint32_t)j;
    }
    gen_sync();
    return (int)g.misc[0];
}

/* Ascore of `ref` against `oth` at `depth` (Ascore.cpp:177-209): per ion type both fragment lists, sorted, the greedy
 * walk for the site-determining ions (ModifiedPeptide.cpp:259-320), their matches of rank <= depth, the two binomial
 * scores from the table.  The value is lane 0's; returns non-zero (wave-uniform) when a list or a trial count is
 * beyond what the launch / the score table was sized for. */
DEV int gen_ascore_pair(const BatchDev &b, const DevConfig *cfg, const GenLds &g, uint64_t ref_bits, uint64_t oth_bits, int depth,
                        int L, int zmax, uint32_t lc, uint32_t list_cap, const PeakEntry *tab, int R, float *asc_out) {
    const int lane = lane_id();
    const float err = cfg->mz_error;
    const bool half_check = err > 0.49f;
    const int T = cfg->n_types, n_fwd = cfg->n_fwd;
    const uint64_t types64 = load_types64(cfg);
    int fail = 0;
    uint32_t tr0 = 0, tr1 = 0, c0 = 0, c1 = 0;              /* (lane 0 keeps the tallies) */
    int tables_dir = -1;
    uint32_t npairs_a = 0, npairs_b = 0;
    for (int t = 0; t < T; t++) {
        const int dir = t < n_fwd ? 0 : 1;
        if (dir != tables_dir) {
            gen_sync();
            uint32_t n = 0;
            if (lane < 2) n = gen_prefix_table(g, cfg, lane ? oth_bits : ref_bits, L, dir, lane, lc);
            npairs_a = (uint32_t)__shfl((int)n, 0, 64);
            npairs_b = (uint32_t)__shfl((int)n, 1, 64);
            tables_dir = dir;
            gen_sync();
        }
        const int na = (int)npairs_a * zmax, nb = (int)npairs_b * zmax;
        if ((uint32_t)na > list_cap || (uint32_t)nb > list_cap) {
            fail = 1;                                       /* (the host sized list_cap for the longest list: not reached) */
            break;
        }
        double A, B;
        type_constants(type_at(types64, t), &A, &B);
        gen_fill_list(g, cfg, L, zmax, 0, lc, A, B, g.la);
        gen_fill_list(g, cfg, L, zmax, 1, lc, A, B, g.lb);
        gen_sync();
        gen_rank_sort(g.la, g.sa, na);
        gen_rank_sort(g.lb, g.sb, nb);
        gen_sync();
        for (int i = lane; i < na; i += 64) g.ha[i] = gen_match_rank(tab, R, g.sa[i], err, half_check) <= depth ? 1 : 0;
        for (int i = lane; i < nb; i += 64) g.hb[i] = gen_match_rank(tab, R, g.sb[i], err, half_check) <= depth ? 1 : 0;
        gen_sync();
        if (lane == 0) {                                    /* the greedy walk (ModifiedPeptide.cpp:291-316) */
            int ia = 0, ib = 0;
            while (ia < na || ib < nb) {
                if (ib == nb) {
                    tr0++;
                    c0 += g.ha[ia++];
                } else if (ia == na) {
                    tr1++;
                    c1 += g.hb[ib++];
                } else {
                    const float xa = g.sa[ia], xb = g.sb[ib];
                    if (__builtin_fabsf(xa - xb) < err) {
                        ia++;
                        ib++;
                    } else if (xa < xb) {
                        tr0++;
                        c0 += g.ha[ia++];
                    } else {
                        tr1++;
                        c1 += g.hb[ib++];
                    }
                }
            }
        }
        gen_sync();
    }
    if (lane == 0 && !fail) {
        if (tr0 > b.lut_n_max || tr1 > b.lut_n_max) {
            fail = 1;
        } else {
            const float sc0 = b.lut[b.lut_off[tr0] + (uint32_t)depth * (tr0 + 1) + c0];
            const float sc1 = b.lut[b.lut_off[tr1] + (uint32_t)depth * (tr1 + 1) + c1];
            *asc_out = sc0 - sc1;
        }
    }
    gen_sync();
    return __any(fail) ? 1 : 0;
}

/* scratch_off[i]: where the i-th PSM of the list has its slice of `scratch` -- sized for ITS OWN number of site assignments
 * and competitors (pya_general_scratch_bytes), not for the launch's largest (r04 advisor finding: one PSM with millions
 * of site assignments made every general PSM of the batch reserve as much) */
__global__ __launch_bounds__(64) void pya_general_psm_kernel(BatchDev b, const uint32_t *ids, uint32_t n_ids, unsigned char *scratch,
                                                              const uint64_t *scratch_off, uint32_t l_cap, uint32_t list_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = ids[blockIdx.x];
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const GenLds g = gen_carve(lds_raw, l_cap, list_cap);
    const uint32_t lc = (l_cap + 3u) & ~3u;
    unsigned char *my = scratch + scratch_off[blockIdx.x];

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
        return;
    }
    const int64_t pep0 = b.pep_off[psm];
    const int L = (int)(b.pep_off[psm + 1] - pep0);
    const int k = b.n_of_mod[psm];
    const int zmax = b.max_charge[psm];
    const int N = (int)b.n_sig[psm];
    PushedEntry *pushed = (PushedEntry *)(my + ((sort_global_bytes((size_t)N) + 15) & ~(size_t)15));
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const uint32_t *inv = b.inv_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];
    const PeakEntry *tab = b.ret + b.ret_off[psm];
    const int R = (int)b.ret_n[psm];
    const float err = cfg->mz_error;
    const bool half_check = err > 0.49f;
    const int T = cfg->n_types, n_fwd = cfg->n_fwd;
    const uint64_t types64 = load_types64(cfg);
    const int ntop = cfg->n_top;                               /* 10..PYA_NTOP_MAX */
    const uint32_t rec_words = (uint32_t)(ntop + 1) / 2u + 1u;    /* count record: ntop 16-bit counts + the fragment total (host_internal.h: rec_words) */

    const int n_sites = gen_setup_residues(b, cfg, g, psm, pep0, L);
    /* (its own competitors: k (n - k) single moves -- what the host sized the slice for) */
    const uint32_t push_cap = k >= 0 && k <= n_sites ? (((uint32_t)k * (uint32_t)(n_sites - k)) + 3u) & ~3u : 0u;

    /* ---- counts and PepScores, one site assignment per lane and trip (Ascore.cpp:53-139) ---- */
    int fail = 0;
    for (int s = lane; s < N; s += 64) {
        const uint64_t bits = order[s];
        uint32_t cnt[PYA_NTOP_MAX];
#pragma unroll
        for (int d = 0; d < PYA_NTOP_MAX; d++) cnt[d] = 0;
        uint32_t nfrag = 0;
        for (int dir = 0; dir < 2; dir++) {
            const int t0 = dir ? n_fwd : 0, t1 = dir ? T : n_fwd;
            if (t0 == t1) continue;
            float running = 0.f;
            uint32_t st = 0;
            uint64_t pm_now = 1ull;
            for (int step = 0; step + 1 < L; step++) {
                const int ri = dir ? L - 1 - step : step;
                const bool mod = gen_modified(g, bits, ri);
                running = (mod ? g.m1[ri] : g.m0[ri]) + running;
                if (cfg->n_nl) {
                    const uint32_t cls = mod ? g.nl1[ri] : g.nl0[ri];
                    if (cls) {
                        const uint32_t st2 = nl_bump(st, cls);
                        if (st2 != st) pm_now = gen_present(g, cfg->n_cand, st2);
                        st = st2;
                    }
                }
                uint64_t pm = pm_now;
                while (pm) {
                    const int v = __builtin_ctzll(pm);
                    pm &= pm - 1;
                    const float x = running - (cfg->n_nl ? g.uniq[v] : 0.f);
                    for (int t = t0; t < t1; t++) {
                        double A, B;
                        type_constants(type_at(types64, t), &A, &B);
                        const double m = ((double)x + A) - B;
                        for (int z = 1; z <= zmax; z++) {
                            const int rk = gen_match_rank(tab, R, charge_mz(m, z), err, half_check);
                            if (rk < ntop) cnt[rk]++;
                            nfrag++;
                        }
                    }
                }
            }
        }
        uint32_t cum[PYA_NTOP_MAX], acc = 0;
#pragma unroll
        for (int d = 0; d < PYA_NTOP_MAX; d++) {
            acc += cnt[d];
            cum[d] = acc;
        }
        float ws = -1.f;
        if (nfrag <= b.lut_n_max) {
            double sum = 0.;
#pragma unroll
            for (int d = 0; d < PYA_NTOP; d++) {
                const float sc = b.lut[b.lut_off[nfrag] + (uint32_t)d * (nfrag + 1) + cum[d]];   /* (the first ten depths are weighted) */
                const float prod = cfg->weights[d] * sc;          /* float product ... */
                sum = sum + (double)prod;                         /* ... double sum    */
            }
            ws = (float)sum;
        } else {
            fail = 1;
        }
        b.ws[s0 + s] = ws;
        uint32_t *r6 = b.rec + (size_t)(s0 + s) * rec_words;
#pragma unroll
        for (int d = 0; d < PYA_NTOP_MAX; d += 2)
            if (d < ntop) r6[d >> 1] = cum[d] | ((d + 1 < ntop ? cum[d + 1] : 0u) << 16);
        r6[rec_words - 1] = nfrag;
    }
    if (__any(fail)) {
        if (lane == 0) {
            b.status[psm] = PYA_ST_LUT_RANGE;
            b.best_score[psm] = -1.f;
            b.best_sig[psm] = 0ull;
            b.n_sig_out[psm] = -1;
        }
        return;
    }
    gen_sync();
    const float *ws = b.ws + s0;

    /* ---- Ascore::isUnambiguous (Ascore.cpp:38-51) ---- */
    if (k >= n_sites) {
        for (int a = lane; a < k && a < (int)max_k; a += 64) out_asc[a] = __builtin_huge_valf();
        if (lane == 0) {
            b.best_score[psm] = N > 0 ? ws[0] : -1.f;
            b.best_sig[psm] = N > 0 ? order[0] : 0ull;
            b.n_sig_out[psm] = N;
            if (b.keep && N > 0) b.sorted_idx[s0] = 0;
        }
        return;
    }

    /* ---- the winner: the front of std::sort (Ascore.cpp:141-146) ---- */
    uint32_t kmax = 0, first_max = 0xffffffffu;
    int n_max = 0;
    for (int i = lane; i < N; i += 64) {
        const uint32_t u = __float_as_uint(ws[i]);             /* scores are >= 0: bit order = value order */
        kmax = u > kmax ? u : kmax;
    }
    kmax = wave_max_u32(kmax);
    for (int i = lane; i < N; i += 64)
        if (__float_as_uint(ws[i]) == kmax) {
            n_max++;
            first_max = first_max < (uint32_t)i ? first_max : (uint32_t)i;
        }
    n_max = wave_sum_i32(n_max);
    first_max = wave_min_u32(first_max);
    uint32_t best_i = first_max;
    if (n_max != 1 || b.keep || (b.debug & 1024)) {
        const SortGlobal srt = sort_carve_global(my, N);
        for (int i = lane; i < N; i += 64) {
            srt.key[i] = ws[i];
            srt.idx[i] = (uint32_t)i;
        }
        srt.sync();
        int front_len = N;
        sort_introsort_loop<false>(srt, N, b.keep == 0, &front_len);
        uint32_t first_pos = 0xffffffffu;
        for (int i = lane; i < front_len; i += 64)
            if (__float_as_uint(srt.key[i]) == kmax) first_pos = first_pos < (uint32_t)i ? first_pos : (uint32_t)i;
        first_pos = wave_min_u32(first_pos);
        best_i = srt.idx[first_pos];
        if (b.keep)
            for (int i = lane; i < N; i += 64) b.sorted_idx[s0 + sort_final_pos(srt, i, N)] = srt.idx[i];
        srt.sync();
    }
    const float best_ws = __uint_as_float(kmax);
    const uint64_t best_bits = order[best_i];

    /* ---- single-move competitors (Ascore.cpp:212-254): the k (n - k) signatures one move away, enumerated through
     * their combination rank; the best of every modified site and its exact ties are kept ---- */
    const uint64_t all_sites = n_sites >= 64 ? ~0ull : ((1ull << n_sites) - 1ull);
    const uint64_t free_bits = all_sites & ~best_bits;
    const int n_free = n_sites - k, items = k * n_free;
    for (int pass = 0; pass < 2; pass++) {
        for (int e = lane; e < items; e += 64) {
            const int a = e / n_free, fb = e - a * n_free;
            const int pos_a = nth_set_bit(best_bits, a), pos_b = nth_set_bit(free_bits, fb);
            const uint64_t c = (best_bits & ~(1ull << pos_a)) | (1ull << pos_b);
            uint32_t rank = 0;
            uint64_t m = c;
            for (int t = 1; m; t++) {                              /* colexicographic rank of the combination */
                const int pos = __builtin_ctzll(m);
                m &= m - 1;
                rank += b.binom[pos * 64 + t];
            }
            const uint32_t idx = inv[rank];
            const uint32_t u = __float_as_uint(ws[idx]);
            if (pass == 0) {
                atomicMax(&g.site_max[a], u);
            } else if (u == g.site_max[a]) {
                if ((double)__builtin_fabsf(best_ws - __uint_as_float(u)) < 1e-6) {
                    g.site_tie[a] = 1u;                            /* ties the winner: Ascore 0 (Ascore.cpp:159-161) */
                    atomicOr(&g.site_alt[a], 1ull << (L <= 64 ? (int)g.site_pos[pos_b] : pos_b));
                } else {
                    const uint32_t slot = atomicAdd(&g.misc[1], 1u);
                    if (slot < push_cap) {
                        PushedEntry pe;
                        pe.bits = c;
                        pe.ws = __uint_as_float(u);
                        pe.idx = idx;
                        pushed[slot] = pe;
                    }
                }
            }
        }
        gen_sync();
    }
    uint32_t np = g.misc[1];
    if (np > push_cap) {                                       /* cannot happen: push_cap >= k (n - k) */
        np = push_cap;
        fail = 2;
    }

    /* ---- Ascores (Ascore.cpp:157-210), one competitor after the other ---- */
    for (uint32_t e = 0; e < np; e++) {
        const PushedEntry pe = pushed[e];
        const uint64_t gone = best_bits & ~pe.bits, came = pe.bits & ~best_bits;
        const int a = __popcll(best_bits & (gone - 1));
        const int came_j = __builtin_ctzll(came);
        if (lane == 0) g.site_alt[a] |= 1ull << (L <= 64 ? (int)g.site_pos[came_j] : came_j);
        /* depth scores of the two from the recorded counts; depth of the largest gap (first one, 0 when none is positive) */
        if (lane < 2 * PYA_NTOP_MAX) {
            const int which = lane / PYA_NTOP_MAX, d = lane - which * PYA_NTOP_MAX;
            if (d < ntop) {
                const uint32_t *r6 = b.rec + (size_t)(s0 + (which ? pe.idx : best_i)) * rec_words;
                const uint32_t cumd = (r6[d >> 1] >> ((d & 1) * 16)) & 0xffffu, nf = r6[rec_words - 1];
                g.sc[lane] = b.lut[b.lut_off[nf] + (uint32_t)d * (nf + 1) + cumd];
            }
        }
        gen_sync();
        int depth = 0;
        {
            float bestd = 0.f;
            for (int d = 0; d < ntop; d++) {
                const float diff = g.sc[d] - g.sc[PYA_NTOP_MAX + d];
                if (diff > bestd) {
                    bestd = diff;
                    depth = d;
                }
            }
        }
        float asc = 0.f;
        if (gen_ascore_pair(b, cfg, g, best_bits, pe.bits, depth, L, zmax, lc, list_cap, tab, R, &asc)) fail = 1;
        else if (lane == 0) g.site_asc[a] = asc < g.site_asc[a] ? asc : g.site_asc[a];
        gen_sync();
    }
    if (lane < k && lane < GEN_MAX_SITES) {
        float asc = g.site_asc[lane];
        if (g.site_tie[lane]) asc = 0.f < asc ? 0.f : asc;
        if (lane < (int)max_k) {
            out_asc[lane] = asc;
            out_alt[lane] = g.site_alt[lane];
        }
    }
    const bool any_fail = __any(fail != 0), overflow = __any(fail == 2);
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (any_fail) b.status[psm] = overflow ? PYA_ST_PUSHED_OVERFLOW : PYA_ST_LUT_RANGE;
    }
}

/* PyAscore.calculate_ambiguity (Ascore.pyx:208-230, Ascore.cpp:157-210) for PSM `psm` of a retained batch with the
 * caller's two score containers, without the fast kernels' limits: any peptide the general kernel takes, n_top up to
 * PYA_NTOP_MAX (`scores` = n_scores depth scores of the reference container, then n_scores of the other), a retained
 * table of any size (looked up where it lies in the workspace).  out[0] = the value, out[1] != 0: a list or a trial
 * count beyond the launch / the score table. */
__global__ __launch_bounds__(64) void pya_general_ambiguity_kernel(BatchDev b, uint32_t psm, uint32_t l_cap, uint32_t list_cap,
                                                                    uint64_t ref_bits, uint64_t oth_bits, const float *scores,
                                                                    uint32_t n_scores, float ref_ws, float oth_ws, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const GenLds g = gen_carve(lds_raw, l_cap, list_cap);
    const uint32_t lc = (l_cap + 3u) & ~3u;
    if ((double)__builtin_fabsf(ref_ws - oth_ws) < 1e-6) {          /* Ascore.cpp:159-161 */
        if (lane == 0) {
            out[0] = 0.f;
            out[1] = 0.f;
        }
        return;
    }
    const int64_t pep0 = b.pep_off[psm];
    const int L = (int)(b.pep_off[psm + 1] - pep0);
    (void)gen_setup_residues(b, cfg, g, psm, pep0, L);
    int depth = 0;
    {
        float bestd = 0.f;                                          /* Ascore.cpp:164-172 */
        for (int d = 0; d < (int)n_scores; d++) {
            const float diff = scores[d] - scores[n_scores + d];
            if (diff > bestd) {
                bestd = diff;
                depth = d;
            }
        }
    }
    float asc = 0.f;
    const int fail = gen_ascore_pair(b, cfg, g, ref_bits, oth_bits, depth, L, b.max_charge[psm], lc, list_cap,
                                     b.ret + b.ret_off[psm], (int)b.ret_n[psm], &asc);
    if (lane == 0) {
        out[0] = asc;
        out[1] = fail ? 1.f : 0.f;
    }
}

extern "C" size_t pya_general_lds_bytes(uint32_t l_cap, uint32_t list_cap) { return gen_lds_bytes(l_cap, list_cap); }
extern "C" size_t pya_general_scratch_bytes(uint32_t n_cap, uint32_t push_cap) {
    return ((sort_global_bytes(n_cap) + 15) & ~(size_t)15) + (size_t)push_cap * sizeof(PushedEntry) + 64;
}

/* d_scratch_off[n_ids]: the PSMs' slices of d_scratch, each pya_general_scratch_bytes(its site assignments, its k (n - k)
 * rounded up to a multiple of 4) long */
extern "C" int pya_launch_general(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch,
                                  const uint64_t *d_scratch_off, uint32_t l_cap, uint32_t list_cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = gen_lds_bytes(l_cap, list_cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_general_psm_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_general_psm_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, d_scratch, d_scratch_off,
                       l_cap, list_cap);
    return (int)hipGetLastError();
}

extern "C" int pya_launch_general_ambiguity(const BatchDev *b, uint32_t psm, uint32_t l_cap, uint32_t list_cap, uint64_t ref_bits,
                                            uint64_t oth_bits, const float *d_scores, uint32_t n_scores, float ref_ws, float oth_ws,
                                            float *d_out, hipStream_t stream) {
    const size_t lds = gen_lds_bytes(l_cap, list_cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_general_ambiguity_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_general_ambiguity_kernel, dim3(1), dim3(64), lds, stream, *b, psm, l_cap, list_cap, ref_bits, oth_bits,
                       d_scores, n_scores, ref_ws, oth_ws, d_out);
    return (int)hipGetLastError();
}
