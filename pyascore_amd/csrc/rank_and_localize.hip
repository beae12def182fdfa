/* rank_and_localize.hip -- kernel 3: best localisation + per-site Ascores (one PSM per wave).
 *
 * Replaces Ascore::sortScores, isUnambiguous, findModifiedPos, calculateAscores,
 * calculateAmbiguity (cpp/Ascore.cpp:38-51, :141-254) and
 * ModifiedPeptide::getSiteDeterminingIons (cpp/ModifiedPeptide.cpp:259-320).
 *
 * The winner among tied PepScores is whatever the reference's std::sort (libstdc++ introsort:
 * median-of-3 Hoare partitions down to 16-element runs, heapsort past the depth limit, then
 * one insertion sort) leaves at the front, so this kernel *emulates that sort exactly*, but
 * wave-parallel: a partition is two ballot/prefix passes that list the scan stops of the two
 * Hoare cursors, pairs them and swaps all crossing pairs at once; the closing insertion sort is
 * a stable sort whose moves never leave a 16-run, so each element's final position is its
 * position plus a count over a +-15 window.  pya_debug_sort exposes it for testing against the
 * host's std::sort.
 *
 * The Ascore of a site compares the site-determining ions of the winner and of its best
 * single-move competitors: fragment lists are generated one prefix length per lane, sorted
 * with an LDS bitonic network, cancelled by the reference's greedy two-pointer walk (serial,
 * lane 0) and matched against the retained-peak table in parallel.
 */
#include "device_common.hip.h"

/* ======================================================================================= */
/* std::sort emulation                                                                      */
/* ======================================================================================= */
struct SortLds {
    float *key;       /* [N] */
    uint16_t *idx;    /* [N] */
    uint16_t *lpos;   /* [N] */
    uint16_t *rpos;   /* [N] */
};

DEV void sort_swap(const SortLds &s, int i, int j) {
    float k = s.key[i];
    s.key[i] = s.key[j];
    s.key[j] = k;
    uint16_t t = s.idx[i];
    s.idx[i] = s.idx[j];
    s.idx[j] = t;
}

/* libstdc++ __adjust_heap / __push_heap with comp(a,b) = key[a] > key[b]; lane 0 only */
DEV void heap_adjust(const SortLds &s, int first, int hole, int len, float vk, uint16_t vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (s.key[first + child] > s.key[first + child - 1]) child--;
        s.key[first + hole] = s.key[first + child];
        s.idx[first + hole] = s.idx[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        s.key[first + hole] = s.key[first + child - 1];
        s.idx[first + hole] = s.idx[first + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && s.key[first + parent] > vk) {
        s.key[first + hole] = s.key[first + parent];
        s.idx[first + hole] = s.idx[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    s.key[first + hole] = vk;
    s.idx[first + hole] = vi;
}

/* __partial_sort(first, last, last) = make_heap + sort_heap; lane 0 only */
DEV void heap_sort_serial(const SortLds &s, int first, int last) {
    int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
        float vk = s.key[first + parent];
        uint16_t vi = s.idx[first + parent];
        heap_adjust(s, first, parent, len, vk, vi);
        if (parent == 0) break;
    }
    while (last - first > 1) {
        --last;
        float vk = s.key[last];
        uint16_t vi = s.idx[last];
        s.key[last] = s.key[first];
        s.idx[last] = s.idx[first];
        heap_adjust(s, first, 0, last - first, vk, vi);
    }
}

/* __unguarded_partition_pivot(first=f, last=l); returns the cut.  Wave-cooperative. */
DEV int sort_partition(const SortLds &s, int f, int l) {
    const int lane = lane_id();
    const int mid = f + (l - f) / 2;
    /* __move_median_to_first(f, f+1, mid, l-1) */
    {
        const float a = s.key[f + 1], b = s.key[mid], c = s.key[l - 1];
        int pick;
        if (a > b) {
            if (b > c) pick = mid;
            else if (a > c) pick = l - 1;
            else pick = f + 1;
        } else if (a > c) pick = f + 1;
        else if (b > c) pick = l - 1;
        else pick = mid;
        wave_lds_sync();
        if (lane == 0) sort_swap(s, f, pick);
        wave_lds_sync();
    }
    const float pv = s.key[f];
    /* stops of the left cursor: positions in [f+1, l) ascending whose key is NOT > pivot */
    int nL = 0;
    for (int base = f + 1; base < l; base += 64) {
        const int i = base + lane;
        const bool stop = i < l && !(s.key[i] > pv);
        const uint64_t m = __ballot(stop);
        if (stop) s.lpos[nL + __popcll(m & lanemask_lt())] = (uint16_t)i;
        nL += __popcll(m);
    }
    /* stops of the right cursor: positions in [f, l) descending for which pivot is NOT > key */
    int nR = 0;
    for (int base = l - 1; base >= f; base -= 64) {
        const int i = base - lane;
        const bool stop = i >= f && !(pv > s.key[i]);
        const uint64_t m = __ballot(stop);
        if (stop) s.rpos[nR + __popcll(m & lanemask_lt())] = (uint16_t)i;
        nR += __popcll(m);
    }
    wave_lds_sync();
    /* pair the r-th stops; they are exchanged while the cursors have not met */
    const int np = nL < nR ? nL : nR;
    int m_sw = 0;
    for (int base = 0; base < np; base += 64) {
        const int r = base + lane;
        bool sw = false;
        int a = 0, b = 0;
        if (r < np) {
            a = s.lpos[r];
            b = s.rpos[r];
            sw = a < b;
        }
        if (sw) sort_swap(s, a, b);
        m_sw += __popcll(__ballot(sw));
    }
    const int cand_l = m_sw < nL ? (int)s.lpos[m_sw] : 0x7fffffff;
    const int cand_r = m_sw >= 1 ? (int)s.rpos[m_sw - 1] : l;
    wave_lds_sync();
    return cand_l < cand_r ? cand_l : cand_r;
}

/* Runs the introsort phase in place.  Afterwards the array is partitioned into runs of <= 16
 * (or heap-sorted runs) exactly as libstdc++ leaves it before __final_insertion_sort. */
DEV void sort_introsort_loop(const SortLds &s, int N) {
    if (N <= 16) return;
    const int lane = lane_id();
    int depth0 = 0;
    for (int t = N; t > 1; t >>= 1) depth0++;
    depth0 *= 2;
    /* explicit stack, one entry per lane */
    int st_f = 0, st_l = 0, st_d = 0;
    int sp = 0;
    if (lane == sp) { st_f = 0; st_l = N; st_d = depth0; }
    sp = 1;
    while (sp > 0) {
        sp--;
        int f = __shfl(st_f, sp, 64), l = __shfl(st_l, sp, 64), d = __shfl(st_d, sp, 64);
        while (l - f > 16) {
            if (d == 0) {
                wave_lds_sync();
                if (lane == 0) heap_sort_serial(s, f, l);
                wave_lds_sync();
                break;
            }
            d--;
            const int cut = sort_partition(s, f, l);
            if (lane == sp) { st_f = cut; st_l = l; st_d = d; }
            sp++;
            l = cut;
        }
    }
    wave_lds_sync();
}

/* final position of element i after the closing (stable) insertion sort */
DEV int sort_final_pos(const SortLds &s, int i, int N) {
    const float me = s.key[i];
    int pos = i;
    const int lo = i - 15 < 0 ? 0 : i - 15;
    const int hi = i + 15 >= N ? N - 1 : i + 15;
    for (int j = lo; j < i; j++) pos -= (me > s.key[j]) ? 1 : 0;     /* moves ahead of smaller keys */
    for (int j = i + 1; j <= hi; j++) pos += (s.key[j] > me) ? 1 : 0;
    return pos;
}

/* ======================================================================================= */
/* signature -> fragments, one prefix length per lane                                       */
/* ======================================================================================= */
struct NlTables {
    const uint16_t *present;  /* LDS [256] */
    const float *uniq;        /* LDS [PYA_MAX_UNIQ] */
    int n_nl;
};

struct Prefix {
    float running;            /* float32 running sum of the lane's prefix                    */
    uint32_t pm;              /* bit set of neutral-loss sums that exist for the prefix      */
};

/* lane i <-> fragment of i+1 residues in direction `dir` of the signature `resmask` */
DEV Prefix prefix_state(const Residues &res, uint64_t resmask, int dir, const NlTables &nl) {
    const int lane = lane_id();
    const int L = res.L;
    float running = 0.f;
    uint32_t st = 0;
    for (int step = 0; step + 1 < L; step++) {
        const int ri = dir == 0 ? step : L - 1 - step;
        const float m0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m0), ri));
        const float m1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(res.m1), ri));
        const uint32_t nlp = (uint32_t)__builtin_amdgcn_readlane((int)res.nl, ri);
        const bool mod = (resmask >> ri) & 1ull;
        const float r = mod ? m1 : m0;
        const uint32_t cls = mod ? (nlp >> 4) : (nlp & 15u);
        if (step <= lane) {
            running = step == 0 ? r : r + running;
            if (cls) st = nl_bump(st, cls);
        }
    }
    Prefix p;
    p.running = running;
    p.pm = (lane + 1 < L) ? (nl.n_nl ? (uint32_t)nl.present[st & 255u] : 1u) : 0u;
    return p;
}

/* PepScore ingredients of one signature: cumulative counts and total fragments (wave-uniform) */
DEV void signature_counts(const Residues &res, const DevConfig *cfg, const NlTables &nl,
                          const PeakTable &tab, uint64_t resmask, int zmax, uint32_t cum[PYA_NTOP],
                          uint32_t *nfrag_out) {
    Hist h = {0ull, 0ull, 0ull};
    int nfrag = 0;
    for (int dir = 0; dir < 2; dir++) {
        const int t0 = dir == 0 ? 0 : cfg->n_fwd;
        const int t1 = dir == 0 ? cfg->n_fwd : cfg->n_types;
        if (t0 == t1) continue;
        Prefix p = prefix_state(res, resmask, dir, nl);
        uint32_t pm = p.pm;
        while (__any(pm != 0)) {
            const bool on = pm != 0;
            const int v = on ? __builtin_ctz(pm) : 0;
            pm &= pm - 1;
            const float x = p.running - (nl.n_nl ? nl.uniq[v] : 0.f);
            const double xd = (double)x;
            for (int t = t0; t < t1; t++) {
                const double m = type_offset(xd, cfg->types[t]);
                for (int z = 1; z <= zmax; z++) {
                    const float fmz = charge_mz(m, z);
                    if (on) {
                        hist_add(h, match_rank(tab, fmz));
                        nfrag++;
                    }
                }
            }
        }
    }
    h = hist_wave_sum(h);
    *nfrag_out = (uint32_t)wave_sum_i32(nfrag);
    uint32_t acc = 0;
#pragma unroll
    for (int d = 0; d < PYA_NTOP; d++) {
        acc += hist_get(h, d);
        cum[d] = acc;
    }
}

/* all fragments of (signature, type) over charges 1..zmax into list[]; returns the count */
DEV int fragment_list(const Residues &res, const DevConfig *cfg, const NlTables &nl, uint64_t resmask,
                      uint8_t type, int zmax, float *list) {
    const int dir = (type == 'b' || type == 'c') ? 0 : 1;
    Prefix p = prefix_state(res, resmask, dir, nl);
    const int mine = __popc(p.pm) * zmax;
    int total;
    int off = wave_excl_scan_i32(mine, &total);
    uint32_t pm = p.pm;
    while (pm) {
        const int v = __builtin_ctz(pm);
        pm &= pm - 1;
        const float x = p.running - (nl.n_nl ? nl.uniq[v] : 0.f);
        const double m = type_offset((double)x, type);
        for (int z = 1; z <= zmax; z++) list[off++] = charge_mz(m, z);
    }
    return total;
}

/* ascending bitonic sort of list[0..n) in LDS; list has room for the next power of two */
DEV int bitonic_sort(float *list, int n) {
    const int lane = lane_id();
    int p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (int i = n + lane; i < p2; i += 64) list[i] = __builtin_huge_valf();
    wave_lds_sync();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (p2 >> 1); t += 64) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const float a = list[lo], b = list[hi];
                if ((a > b) == up) {
                    list[lo] = b;
                    list[hi] = a;
                }
            }
            wave_lds_sync();
        }
    }
    return p2;
}

struct AmbLds {
    float *la;          /* [list_cap] */
    float *lb;          /* [list_cap] */
    uint8_t *ka;        /* [list_cap] */
    uint8_t *kb;        /* [list_cap] */
};

/* Ascore::calculateAmbiguity (cpp/Ascore.cpp:157-210).  Scores are passed in (wave-uniform). */
DEV float ambiguity(const BatchDev &b, const Residues &res, const DevConfig *cfg, const NlTables &nl,
                    const PeakTable &tab, const AmbLds &w, int zmax, uint64_t ref_mask,
                    const float ref_scores[PYA_NTOP], float ref_ws, uint64_t oth_mask,
                    const float oth_scores[PYA_NTOP], float oth_ws, int *fail) {
    const int lane = lane_id();
    if ((double)__builtin_fabsf(ref_ws - oth_ws) < 1e-6) return 0.f;
    float best = 0.f;
    int depth = 0;
#pragma unroll
    for (int d = 0; d < PYA_NTOP; d++) {
        const float diff = ref_scores[d] - oth_scores[d];
        if (diff > best) {
            best = diff;
            depth = d;
        }
    }
    int cnt0 = 0, cnt1 = 0, tr0 = 0, tr1 = 0;
    const float err = cfg->mz_error;
    for (int t = 0; t < cfg->n_types; t++) {
        const uint8_t type = cfg->types[t];
        wave_lds_sync();
        const int na = fragment_list(res, cfg, nl, ref_mask, type, zmax, w.la);
        const int nb = fragment_list(res, cfg, nl, oth_mask, type, zmax, w.lb);
        wave_lds_sync();
        bitonic_sort(w.la, na);
        bitonic_sort(w.lb, nb);
        /* greedy cancellation, ModifiedPeptide.cpp:291-316 */
        if (lane == 0) {
            int i = 0, j = 0;
            while (i < na || j < nb) {
                if (j == nb) {
                    w.ka[i++] = 1;
                } else if (i == na) {
                    w.kb[j++] = 1;
                } else {
                    const float x = w.la[i], y = w.lb[j];
                    if (__builtin_fabsf(x - y) < err) {
                        w.ka[i++] = 0;
                        w.kb[j++] = 0;
                    } else if (x < y) {
                        w.ka[i++] = 1;
                    } else {
                        w.kb[j++] = 1;
                    }
                }
            }
        }
        wave_lds_sync();
        for (int i = lane; i < na; i += 64) {
            if (w.ka[i]) {
                tr0++;
                cnt0 += match_rank(tab, w.la[i]) <= depth ? 1 : 0;
            }
        }
        for (int j = lane; j < nb; j += 64) {
            if (w.kb[j]) {
                tr1++;
                cnt1 += match_rank(tab, w.lb[j]) <= depth ? 1 : 0;
            }
        }
    }
    cnt0 = wave_sum_i32(cnt0);
    cnt1 = wave_sum_i32(cnt1);
    tr0 = wave_sum_i32(tr0);
    tr1 = wave_sum_i32(tr1);
    if ((uint32_t)tr0 > b.lut_n_max || (uint32_t)tr1 > b.lut_n_max) {
        *fail = 1;
        return 0.f;
    }
    const float s0 = b.lut[b.lut_off[tr0] + (uint32_t)depth * (uint32_t)(tr0 + 1) + (uint32_t)cnt0];
    const float s1 = b.lut[b.lut_off[tr1] + (uint32_t)depth * (uint32_t)(tr1 + 1) + (uint32_t)cnt1];
    return s0 - s1;
}

DEV void scores_from_counts(const BatchDev &b, const uint32_t cum[PYA_NTOP], uint32_t nfrag,
                            float out[PYA_NTOP], int *fail) {
    if (nfrag > b.lut_n_max) {
        *fail = 1;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d++) out[d] = 0.f;
        return;
    }
    const uint32_t off = b.lut_off[nfrag];
#pragma unroll
    for (int d = 0; d < PYA_NTOP; d++) out[d] = b.lut[off + (uint32_t)d * (nfrag + 1) + cum[d]];
}

DEV int nth_set_bit(uint64_t m, int n) {
    for (int i = 0; i < n; i++) m &= m - 1;
    return __builtin_ctzll(m);
}

/* LDS carve-up shared by the localisation and the ambiguity kernels */
struct K3Lds {
    uint16_t *nl_present;
    float *nl_uniq;
    float *t_mz;
    uint8_t *t_rank;
    uint32_t *pushed;        /* [PYA_MAX_PUSHED] */
    uint32_t *site_max;      /* [64] */
    uint32_t *n_pushed;      /* [1]  */
    unsigned char *scratch;  /* sort arrays, later the ambiguity lists */
};

DEV K3Lds carve(unsigned char *raw, uint32_t peak_cap) {
    K3Lds k;
    k.nl_present = (uint16_t *)raw;
    k.nl_uniq = (float *)(k.nl_present + 256);
    k.pushed = (uint32_t *)(k.nl_uniq + PYA_MAX_UNIQ);
    k.site_max = k.pushed + PYA_MAX_PUSHED;
    k.n_pushed = k.site_max + 64;
    k.t_mz = (float *)(k.n_pushed + 4);
    k.t_rank = (uint8_t *)(k.t_mz + peak_cap);
    k.scratch = (unsigned char *)(k.t_rank + ((peak_cap + 15u) & ~15u));
    return k;
}

extern "C" size_t pya_localize_lds_bytes(uint32_t peak_cap, uint32_t n_cap, uint32_t list_cap) {
    size_t fixed = 512 + PYA_MAX_UNIQ * 4 + PYA_MAX_PUSHED * 4 + 64 * 4 + 16 + (size_t)peak_cap * 4 +
                   ((peak_cap + 15u) & ~15u);
    size_t srt = (size_t)n_cap * 10 + 64;
    size_t lst = (size_t)list_cap * 10 + 64;
    return fixed + (srt > lst ? srt : lst) + 64;
}

DEV void stage_tables(const BatchDev &b, const DevConfig *cfg, const K3Lds &k, uint32_t psm,
                      PeakTable *tab, NlTables *nl) {
    const int lane = lane_id();
    const int64_t p0 = b.peak_off[psm];
    const int R = (int)b.ret_n[psm];
    for (int i = lane; i < R; i += 64) {
        k.t_mz[i] = b.ret_mz[p0 + i];
        k.t_rank[i] = b.ret_rank[p0 + i];
    }
    nl->n_nl = cfg->n_nl;
    if (nl->n_nl) {
        for (int i = lane; i < 256; i += 64) k.nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) k.nl_uniq[lane] = cfg->uniq[lane];
    }
    nl->present = k.nl_present;
    nl->uniq = k.nl_uniq;
    tab->mz = k.t_mz;
    tab->rank = k.t_rank;
    tab->n = R;
    tab->pow2 = 1;
    while (tab->pow2 < R) tab->pow2 <<= 1;
    tab->err = cfg->mz_error;
}

__global__ __launch_bounds__(64) void pya_localize_kernel(BatchDev b, const uint32_t *psm_ids,
                                                          uint32_t n_ids, uint32_t peak_cap,
                                                          uint32_t list_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[blockIdx.x];
    const int lane = lane_id();
    const DevConfig *cfg = b.cfg;
    const int k = b.n_of_mod[psm];
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

    const int N = (int)b.n_sig[psm];
    const int n_sites = (int)b.n_sites[psm];
    const uint64_t *order = b.order_tab + b.order_off[psm];
    const int64_t s0 = b.sig_off[psm];
    const float *ws = b.ws + s0;

    /* Ascore::isUnambiguous, cpp/Ascore.cpp:38-51 */
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

    K3Lds lds = carve(lds_raw, peak_cap);
    PeakTable tab;
    NlTables nl;
    stage_tables(b, cfg, lds, psm, &tab, &nl);

    /* ---- sort (cpp/Ascore.cpp:141-146) ---- */
    SortLds srt;
    srt.key = (float *)lds.scratch;
    srt.idx = (uint16_t *)(srt.key + N);
    srt.lpos = srt.idx + N;
    srt.rpos = srt.lpos + N;
    for (int i = lane; i < N; i += 64) {
        srt.key[i] = ws[i];
        srt.idx[i] = (uint16_t)i;
    }
    if (lane == 0) *lds.n_pushed = 0;
    lds.site_max[lane] = 0;
    wave_lds_sync();
    sort_introsort_loop(srt, N);

    /* front of the sorted list = left-most maximum of the partitioned array */
    uint32_t kmax = 0;
    for (int i = lane; i < N; i += 64) {
        uint32_t u = __float_as_uint(srt.key[i]);          /* scores are >= 0: bit order = value order */
        kmax = u > kmax ? u : kmax;
    }
    kmax = wave_max_u32(kmax);
    uint32_t first_pos = 0xffffffffu;
    for (int i = lane; i < N; i += 64)
        if (__float_as_uint(srt.key[i]) == kmax) first_pos = first_pos < (uint32_t)i ? first_pos : (uint32_t)i;
    first_pos = wave_min_u32(first_pos);
    const uint32_t best_i = srt.idx[first_pos];
    const float best_ws = __uint_as_float(kmax);
    const uint64_t best_bits = order[best_i];
    if (b.keep) {
        for (int i = lane; i < N; i += 64) b.sorted_idx[s0 + sort_final_pos(srt, i, N)] = srt.idx[i];
    }
    wave_lds_sync();

    /* ---- single-move competitors (cpp/Ascore.cpp:212-254) ---- */
    for (int pass = 0; pass < 2; pass++) {
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            if (i < N) {
                const uint64_t c = order[i];
                const uint64_t gone = best_bits & ~c, came = c & ~best_bits;
                if (__popcll(gone) == 1 && __popcll(came) == 1) {
                    const int a = __popcll(best_bits & (gone - 1));
                    const uint32_t u = __float_as_uint(ws[i]);
                    if (pass == 0) {
                        atomicMax(&lds.site_max[a], u);
                    } else if (u == lds.site_max[a]) {
                        const uint32_t slot = atomicAdd(lds.n_pushed, 1u);
                        if (slot < PYA_MAX_PUSHED) lds.pushed[slot] = (uint32_t)i;
                    }
                }
            }
        }
        wave_lds_sync();
    }
    const uint32_t n_pushed = *lds.n_pushed;
    int fail = 0;
    if (n_pushed > PYA_MAX_PUSHED) fail = 2;

    const Residues res = load_residues(b, cfg, psm);
    const int zmax = b.max_charge[psm];
    const uint64_t best_mask = deposit_sites(best_bits, res.site_mask);
    uint32_t cum[PYA_NTOP], nfrag;
    float best_scores[PYA_NTOP];
    signature_counts(res, cfg, nl, tab, best_mask, zmax, cum, &nfrag);
    scores_from_counts(b, cum, nfrag, best_scores, &fail);

    AmbLds amb;
    amb.la = (float *)lds.scratch;
    amb.lb = amb.la + list_cap;
    amb.ka = (uint8_t *)(amb.lb + list_cap);
    amb.kb = amb.ka + list_cap;

    /* per modified site: min Ascore over the tied best competitors, their positions as a mask */
    float my_asc = __builtin_huge_valf();     /* lane a keeps site a */
    uint64_t my_alt = 0ull;
    const uint32_t np = n_pushed < PYA_MAX_PUSHED ? n_pushed : PYA_MAX_PUSHED;
    for (uint32_t e = 0; e < np; e++) {
        const uint32_t ci = lds.pushed[e];
        const uint64_t c = order[ci];
        const float c_ws = ws[ci];
        const uint64_t gone = best_bits & ~c, came = c & ~best_bits;
        const int a = __popcll(best_bits & (gone - 1));
        const int q = __builtin_ctzll(came);
        float asc = 0.f;
        if (!((double)__builtin_fabsf(best_ws - c_ws) < 1e-6)) {
            const uint64_t c_mask = deposit_sites(c, res.site_mask);
            float c_scores[PYA_NTOP];
            signature_counts(res, cfg, nl, tab, c_mask, zmax, cum, &nfrag);
            scores_from_counts(b, cum, nfrag, c_scores, &fail);
            asc = ambiguity(b, res, cfg, nl, tab, amb, zmax, best_mask, best_scores, best_ws, c_mask,
                            c_scores, c_ws, &fail);
        }
        if (lane == a) {
            my_asc = asc < my_asc ? asc : my_asc;
            my_alt |= 1ull << nth_set_bit(res.site_mask, q);
        }
    }
    if (lane < k && lane < (int)max_k) {
        out_asc[lane] = my_asc;
        out_alt[lane] = my_alt;
    }
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (fail) b.status[psm] = fail == 2 ? PYA_ST_PUSHED_OVERFLOW : PYA_ST_LUT_RANGE;
    }
}

/* PyAscore.calculate_ambiguity for PSM `psm` with caller-supplied score containers */
__global__ __launch_bounds__(64) void pya_ambiguity_kernel(BatchDev b, uint32_t psm, uint32_t peak_cap,
                                                           uint32_t list_cap, uint64_t ref_bits,
                                                           uint64_t oth_bits, const float *scores,
                                                           float ref_ws, float oth_ws, float *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const DevConfig *cfg = b.cfg;
    K3Lds lds = carve(lds_raw, peak_cap);
    PeakTable tab;
    NlTables nl;
    stage_tables(b, cfg, lds, psm, &tab, &nl);
    wave_lds_sync();
    const Residues res = load_residues(b, cfg, psm);
    AmbLds amb;
    amb.la = (float *)lds.scratch;
    amb.lb = amb.la + list_cap;
    amb.ka = (uint8_t *)(amb.lb + list_cap);
    amb.kb = amb.ka + list_cap;
    float rs[PYA_NTOP], os[PYA_NTOP];
#pragma unroll
    for (int d = 0; d < PYA_NTOP; d++) {
        rs[d] = scores[d];
        os[d] = scores[PYA_NTOP + d];
    }
    int fail = 0;
    const float v = ambiguity(b, res, cfg, nl, tab, amb, b.max_charge[psm],
                              deposit_sites(ref_bits, res.site_mask), rs, ref_ws,
                              deposit_sites(oth_bits, res.site_mask), os, oth_ws, &fail);
    if (lane_id() == 0) {
        out[0] = v;
        out[1] = fail ? 1.f : 0.f;
    }
}

/* test hook: the sort alone, on caller keys */
__global__ __launch_bounds__(64) void pya_debug_sort_kernel(const float *keys, uint32_t n, uint32_t *perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int N = (int)n;
    SortLds srt;
    srt.key = (float *)lds_raw;
    srt.idx = (uint16_t *)(srt.key + N);
    srt.lpos = srt.idx + N;
    srt.rpos = srt.lpos + N;
    for (int i = lane; i < N; i += 64) {
        srt.key[i] = keys[i];
        srt.idx[i] = (uint16_t)i;
    }
    wave_lds_sync();
    sort_introsort_loop(srt, N);
    for (int i = lane; i < N; i += 64) perm[sort_final_pos(srt, i, N)] = srt.idx[i];
}

extern "C" int pya_launch_localize(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids,
                                   uint32_t peak_cap, uint32_t n_cap, uint32_t list_cap,
                                   hipStream_t stream) {
    if (n_ids == 0) return 0;
    size_t lds = pya_localize_lds_bytes(peak_cap, n_cap, list_cap);
    hipError_t e = hipFuncSetAttribute((const void *)pya_localize_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_localize_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids,
                       peak_cap, list_cap);
    return (int)hipGetLastError();
}

extern "C" int pya_launch_ambiguity(const BatchDev *b, uint32_t psm, uint32_t peak_cap, uint32_t list_cap,
                                    uint64_t ref_bits, uint64_t oth_bits, const float *d_scores,
                                    float ref_ws, float oth_ws, float *d_out, hipStream_t stream) {
    size_t lds = pya_localize_lds_bytes(peak_cap, 0, list_cap);
    hipError_t e = hipFuncSetAttribute((const void *)pya_ambiguity_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_ambiguity_kernel, dim3(1), dim3(64), lds, stream, *b, psm, peak_cap, list_cap,
                       ref_bits, oth_bits, d_scores, ref_ws, oth_ws, d_out);
    return (int)hipGetLastError();
}

extern "C" int pya_launch_debug_sort(const float *d_keys, uint32_t n, uint32_t *d_perm, hipStream_t stream) {
    size_t lds = (size_t)n * 10 + 64;
    hipError_t e = hipFuncSetAttribute((const void *)pya_debug_sort_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_debug_sort_kernel, dim3(1), dim3(64), lds, stream, d_keys, n, d_perm);
    return (int)hipGetLastError();
}
