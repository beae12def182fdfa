/* localize_core.hip.h -- device code of rank_and_localize.hip:
 * the std::sort emulation, the per-signature fragment machinery and the batched Ascore
 * computation.  See rank_and_localize.hip for the algorithm notes and reference citations. */
#ifndef PYA_LOCALIZE_CORE_H
#define PYA_LOCALIZE_CORE_H
#include "device_common.hip.h"


/* ======================================================================================= */
/* std::sort emulation                                                                      */
/* ======================================================================================= */
/* Work area of the sort emulation: 6 bytes per element (key, index) plus, per 64 elements, the two
 * 64-bit masks of a partition pass's cursor stops, plus two 128-entry queues the stops are unpacked
 * into as the pass pairs them.  (Listing every stop -- 4 more bytes per element -- made the sort room
 * 10 bytes per signature, and that room decides how many wavefronts a CU holds while thousands of
 * scores are sorted: 30 KB -> 19 KB for 3003.) */
template <typename IdxT, bool GLOBAL>
struct SortAreaT {
    typedef IdxT idx_t;
    float *key;          /* [N] */
    IdxT *idx;           /* [N] */
    uint64_t *lmask;     /* [N / 64 + 2] stops of the left cursor, chunk by chunk from the left   */
    uint64_t *rmask;     /* [N / 64 + 2] stops of the right cursor, chunk by chunk from the right */
    IdxT *lq, *rq;       /* [qmask + 1] each: positions of the stops being paired                */
    int qmask;           /* 127; 63 when there are at most 64 elements (no stop index reaches 64) */
    /* lanes hand data to each other through the arrays: LDS (the kernels' own sorts), or the workspace in global
     * memory (general_psm.hip: any number of site assignments, 32-bit positions) */
    DEV void sync() const {
        if (GLOBAL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            wave_lds_sync();
        }
    }
};
typedef SortAreaT<uint16_t, false> SortLds;
typedef SortAreaT<uint32_t, true> SortGlobal;
/* work area of a sort of N elements in global memory: key f32[N] | idx u32[N] | masks u64[2 (N / 64 + 2)] | queues u32[2][128] */
__host__ __device__ static inline size_t sort_global_bytes(size_t n) {
    return ((n * 8 + 15) & ~(size_t)15) + (n / 64 + 2) * 16 + 2 * 128 * 4;
}
DEV SortGlobal sort_carve_global(unsigned char *raw, int N) {
    SortGlobal s;
    s.key = (float *)raw;
    s.idx = (uint32_t *)(s.key + N);
    unsigned char *m = raw + (((size_t)N * 8 + 15) & ~(size_t)15);
    s.lmask = (uint64_t *)m;
    s.rmask = s.lmask + (N / 64 + 2);
    s.lq = (uint32_t *)(s.rmask + (N / 64 + 2));
    s.rq = s.lq + 128;
    s.qmask = N <= 64 ? 63 : 127;
    return s;
}
__host__ __device__ static inline size_t sort_lds_bytes(size_t n) {
    if (n <= 64) return (n * 10 + 15) & ~(size_t)15;      /* two plain lists of n stops, no masks */
    return ((n * 6 + 15) & ~(size_t)15) + (n / 64 + 2) * 16 + 2 * 128 * 2;
}
DEV SortLds sort_carve(unsigned char *raw, int N) {
    SortLds s;
    s.key = (float *)raw;
    s.idx = (uint16_t *)(s.key + N);
    if (N <= 64) {
        s.qmask = 63;
        s.lmask = s.rmask = nullptr;
        s.lq = s.idx + N;
        s.rq = s.lq + N;
        return s;
    }
    unsigned char *m = raw + (((size_t)N * 6 + 15) & ~(size_t)15);
    s.lmask = (uint64_t *)m;
    s.rmask = s.lmask + (N / 64 + 2);
    s.qmask = 127;
    s.lq = (uint16_t *)(s.rmask + (N / 64 + 2));
    s.rq = s.lq + 128;
    return s;
}

template <class S>
DEV void sort_swap(const S &s, int i, int j) {
    float k = s.key[i];
    s.key[i] = s.key[j];
    s.key[j] = k;
    typename S::idx_t t = s.idx[i];
    s.idx[i] = s.idx[j];
    s.idx[j] = t;
}

/* libstdc++ __adjust_heap / __push_heap with comp(a,b) = key[a] > key[b]; lane 0 only */
template <class S>
DEV void heap_adjust(const S &s, int first, int hole, int len, float vk, typename S::idx_t vi) {
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
template <class S>
DEV void heap_sort_serial(const S &s, int first, int last) {
    int len = last - first;
    if (len < 2) return;
    for (int parent = (len - 2) / 2;; parent--) {
        float vk = s.key[first + parent];
        typename S::idx_t vi = s.idx[first + parent];
        heap_adjust(s, first, parent, len, vk, vi);
        if (parent == 0) break;
    }
    while (last - first > 1) {
        --last;
        float vk = s.key[last];
        typename S::idx_t vi = s.idx[last];
        s.key[last] = s.key[first];
        s.idx[last] = s.idx[first];
        heap_adjust(s, first, 0, last - first, vk, vi);
    }
}

/* __unguarded_partition_pivot(first=f, last=l); returns the cut.  Wave-cooperative.
 * LEFT_ONLY: the caller follows the left part only (the front of the sorted list is all it wants), so
 * the elements a swap would move into the right part are not written.
 * A wavefront here has little company on its CU (the sort room decides the occupancy), so the sweeps
 * read four chunks before they use the first: one LDS round trip per 256 elements, not per 64. */
/* SMALL_ONLY: the caller never has more than 64 elements (the fused kernel): the general path is not
 * compiled in -- its arrays would cost that kernel a scratch allocation. */
template <bool LEFT_ONLY, bool SMALL_ONLY = false, class S>
DEV int sort_partition(const S &s, int f, int l) {
    typedef typename S::idx_t idx_t;
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
        s.sync();
        if (lane == 0) sort_swap(s, f, pick);
        s.sync();
    }
    const float pv = s.key[f];
    if (SMALL_ONLY || s.qmask == 63) {
        /* at most 64 elements: one chunk per cursor, and every stop fits the queues, which then are
         * plain lists */
        const int il = f + 1 + lane, ir = l - 1 - lane;
        const float kl = s.key[il < l ? il : l - 1], kr = s.key[ir >= f ? ir : f];
        const bool stop_l = il < l && !(kl > pv), stop_r = ir >= f && !(pv > kr);
        const uint64_t ml = __ballot(stop_l), mr = __ballot(stop_r);
        if (stop_l) s.lq[mask_rank(ml)] = (idx_t)il;
        if (stop_r) s.rq[mask_rank(mr)] = (idx_t)ir;
        const int nL = __popcll(ml), nR = __popcll(mr);
        s.sync();
        const int np = nL < nR ? nL : nR;
        int a = 0, b = 0;
        if (lane < np) {
            a = s.lq[lane];
            b = s.rq[lane];
        }
        const bool sw = lane < np && a < b;
        const float ka = s.key[a], kb = s.key[b];
        const idx_t ia = s.idx[a], ib = s.idx[b];
        if (sw) {
            s.key[a] = kb;
            s.idx[a] = ib;
            if (!LEFT_ONLY) {
                s.key[b] = ka;
                s.idx[b] = ia;
            }
        }
        const int m_sw = __popcll(__ballot(sw));
        const int cand_l = m_sw < nL ? (int)s.lq[m_sw] : 0x7fffffff;
        const int cand_r = m_sw >= 1 ? (int)s.rq[m_sw - 1] : l;
        s.sync();
        return cand_l < cand_r ? cand_l : cand_r;
    }
    if (SMALL_ONLY) return l;                               /* (not reached) */
    /* stops of the left cursor: positions in [f+1, l) ascending whose key is NOT > pivot -- one mask
     * per chunk of 64 positions */
    int nL = 0, nLc = 0;
    for (int base = f + 1; base < l; base += 256) {
        float k[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base + u * 64 + lane;
            k[u] = s.key[i < l ? i : l - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base + u * 64 < l) {                         /* (wave-uniform) */
                const int i = base + u * 64 + lane;
                const uint64_t m = __ballot(i < l && !(k[u] > pv));
                if (lane == 0) s.lmask[nLc] = m;
                nLc++;
                nL += __popcll(m);
            }
        }
    }
    /* stops of the right cursor: positions in [f, l) descending for which pivot is NOT > key */
    int nR = 0, nRc = 0;
    for (int base = l - 1; base >= f; base -= 256) {
        float k[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = base - u * 64 - lane;
            k[u] = s.key[i >= f ? i : f];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (base - u * 64 >= f) {
                const int i = base - u * 64 - lane;
                const uint64_t m = __ballot(i >= f && !(pv > k[u]));
                if (lane == 0) s.rmask[nRc] = m;
                nRc++;
                nR += __popcll(m);
            }
        }
    }
    s.sync();
    /* Pair the r-th stops; they are exchanged while the cursors have not met (the left stops ascend, the
     * right ones descend: once a pair has met, all later ones have).  The stops of a chunk are unpacked
     * into the queues -- entry r at slot r mod 128 (mod 64 for at most 64 elements) -- just ahead of the block of 64 pairs that needs them. */
    const int np = nL < nR ? nL : nR;
    int m_sw = 0, lprod = 0, rprod = 0, lc = 0, rc = 0;
    auto unpack_left = [&](int upto) {                     /* until entry `upto` exists (or no chunk is left) */
        while (lprod <= upto && lc < nLc) {
            const uint64_t mv = s.lmask[lc];
            const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mv >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mv);
            if ((m >> lane) & 1ull) s.lq[(lprod + mask_rank(m)) & s.qmask] = (idx_t)(f + 1 + lc * 64 + lane);
            lprod += __popcll(m);
            lc++;
        }
    };
    auto unpack_right = [&](int upto) {
        while (rprod <= upto && rc < nRc) {
            const uint64_t mv = s.rmask[rc];
            const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mv >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mv);
            if ((m >> lane) & 1ull) s.rq[(rprod + mask_rank(m)) & s.qmask] = (idx_t)(l - 1 - rc * 64 - lane);
            rprod += __popcll(m);
            rc++;
        }
    };
    for (int r0 = 0; r0 < np; r0 += 64) {
        unpack_left(r0 + 63);
        unpack_right(r0 + 63);
        s.sync();
        const int r = r0 + lane;
        int a = 0, b = 0;
        if (r < np) {
            a = s.lq[r & s.qmask];
            b = s.rq[r & s.qmask];
        }
        const bool sw = r < np && a < b;
        const float ka = s.key[a], kb = s.key[b];
        const idx_t ia = s.idx[a], ib = s.idx[b];
        if (sw) {
            s.key[a] = kb;
            s.idx[a] = ib;
            if (!LEFT_ONLY) {
                s.key[b] = ka;
                s.idx[b] = ia;
            }
        }
        const int n_sw = __popcll(__ballot(sw));
        m_sw += n_sw;
        s.sync();
        if (n_sw < 64) break;                              /* the cursors have met (or the pairs ran out) */
    }
    int cand_l = 0x7fffffff;
    if (m_sw < nL) {
        unpack_left(m_sw);
        s.sync();
        cand_l = (int)s.lq[m_sw & s.qmask];
    }
    const int cand_r = m_sw >= 1 ? (int)s.rq[(m_sw - 1) & s.qmask] : l;
    s.sync();
    return cand_l < cand_r ? cand_l : cand_r;
}

/* Runs the introsort phase in place.  Afterwards the array is partitioned into runs of <= 16
 * (or heap-sorted runs) exactly as libstdc++ leaves it before __final_insertion_sort. */
/* PLAIN (the lean instantiation of the localize kernel, see rank_and_localize.hip) gives up --
 * returns true -- where the depth limit would call for the serial heap sort. */
template <bool PLAIN, bool SMALL_ONLY = false, class S>
DEV bool sort_introsort_loop(const S &s, int N, bool spine_only, int *front_len = nullptr) {
    if (front_len) *front_len = N;
    if (N <= 16) return false;
    const int lane = lane_id();
    int depth0 = 0;
    for (int t = N; t > 1; t >>= 1) depth0++;
    depth0 *= 2;
    if (spine_only) {
        /* the front of the sorted list comes out of the left-most run: follow the left parts only.
         * (The discarded right parts keep stale copies of what was swapped out of them.) */
        int l = N, d = depth0;
        while (l > 16) {
            if (d == 0) {
                if (PLAIN) return true;
                s.sync();
                if (lane == 0) heap_sort_serial(s, 0, l);   /* (stale right parts do not reach in here) */
                s.sync();
                break;
            }
            d--;
            l = sort_partition<true, SMALL_ONLY>(s, 0, l);
        }
        if (front_len) *front_len = l;
        s.sync();
        return false;
    }
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
                if (PLAIN) return true;
                s.sync();
                if (lane == 0) heap_sort_serial(s, f, l);
                s.sync();
                break;
            }
            d--;
            const int cut = sort_partition<false, SMALL_ONLY>(s, f, l);
            if (lane == sp) { st_f = cut; st_l = l; st_d = d; }
            sp++;
            l = cut;
        }
    }
    s.sync();
    return false;
}

/* final position of element i after the closing (stable) insertion sort */
template <class S>
DEV int sort_final_pos(const S &s, int i, int N) {
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
    const float s0 = b.lut[lut_row(tr0) + (uint32_t)depth * (uint32_t)(tr0 + 1) + (uint32_t)cnt0];
    const float s1 = b.lut[lut_row(tr1) + (uint32_t)depth * (uint32_t)(tr1 + 1) + (uint32_t)cnt1];
    return s0 - s1;
}

/* LDS carve-up shared by the localisation and the ambiguity kernels */
struct K3Lds {
    uint16_t *nl_present;
    float *nl_uniq;
    PeakEntry *t_e;          /* [peak_cap + PYA_TABLE_PAD] */
    PushedEntry *pushed;     /* [push_cap]: at most k * (n_sites - k) single-move competitors exist */
    uint32_t *site_max;      /* [64] best competitor PepScore (bits) per modified site */
    uint32_t *site_tie;      /* [64] some best competitor ties the winner (Ascore 0)   */
    unsigned long long *site_alt;  /* [64] positions of competitors that tie the winner */
    uint32_t *n_pushed;      /* [4]  */
    uint16_t *grid;          /* [PYA_GRID_CELLS] */
    unsigned char *scratch;  /* sort arrays, later the localisation work area */
    uint32_t nl_cap;         /* entries of nl_present staged (256, or 4^(distinct loss masses) in the hash route) */
};

/* site_cap: entries of the three per-site arrays (64, or the launch's largest number of modified sites rounded up) */
DEV K3Lds carve(unsigned char *raw, uint32_t peak_cap, bool with_table = true, uint32_t push_cap = PYA_MAX_PUSHED,
                uint32_t site_cap = 64, uint32_t nl_cap = 256) {
    K3Lds k;
    k.nl_cap = nl_cap;
    k.nl_present = (uint16_t *)raw;
    k.nl_uniq = (float *)(k.nl_present + nl_cap);
    k.pushed = (PushedEntry *)(k.nl_uniq + PYA_MAX_UNIQ);
    k.site_alt = (unsigned long long *)(k.pushed + push_cap);
    k.site_max = (uint32_t *)(k.site_alt + site_cap);
    k.site_tie = k.site_max + site_cap;
    k.n_pushed = k.site_tie + site_cap;
    k.grid = (uint16_t *)(k.n_pushed + 4);
    k.t_e = (PeakEntry *)(k.grid + (with_table ? PYA_GRID_CELLS : 0));
    k.scratch = (unsigned char *)(k.t_e + (with_table ? peak_cap + PYA_TABLE_PAD : 0));
    return k;
}


/* work area of the batched localisation (aliases the sort arrays) */
struct LocLds {
    float *m0, *m1;           /* [64] residue masses                                      */
    uint8_t *nlp;             /* [64] NL classes                                          */
    uint64_t *sig_mask;       /* [sb] residue masks, entry 0 = winner                 */
    float *run;               /* [sb*2*pos_cap] running sums per (sig, dir, prefix)   */
    uint16_t *pmk;            /* same shape: neutral-loss sums present                    */
    uint16_t *cpre;           /* same shape: exclusive count of variants before prefix    */
    uint32_t *tot;            /* [sb*2] variants per (sig, dir)                        */
    float *scores;            /* [sb*10]                                              */
    uint32_t *c_idx;          /* [sb] which modified site of the winner the competitor moves */
    uint32_t *c_pre;          /* [sb] pre-sort index of the signature (entry 0 = winner)     */
    int32_t *c_depth;         /* [sb]                                                 */
    uint32_t *c_cnt;          /* [sb*2] matched site-determining ions (ref, other)    */
    uint32_t *c_tr;           /* [sb*2] site-determining ions                         */
    float *pool;              /* [pool_cap] fragment lists [slot][type slot][P2]          */
    uint8_t *keep;            /* [2 * pool_cap] keep flags [task][side][P2]               */
    float *stage_val;         /* [LOC_STAGE] surviving ions waiting for their lookup      */
    uint32_t *stage_tag;      /* [LOC_STAGE] competitor * 2 + side                        */
    uint32_t *t_lo, *t_off;   /* [32], [33] per task: first step of its span, items before it (lean pairing) */
};

/* res_cap: entries of the residue arrays (64, or the launch's longest peptide rounded up to a multiple of 4);
 * lean_tables = false leaves the span tables of the lean pairing out (the hash route: LDS decides its occupancy) */
/* nl_tables = false (the lean instantiation, r06): no room for the loss-variant tables pmk / cpre, which only the general
 * settings read -- a quarter of the run tables' bytes */
DEV LocLds loc_carve(unsigned char *raw, uint32_t pos_cap, uint32_t pool_cap, uint32_t LOC_SB, uint32_t res_cap = 64,
                     bool lean_tables = true, bool nl_tables = true) {
    LocLds w;
    w.sig_mask = (uint64_t *)raw;
    w.m0 = (float *)(w.sig_mask + LOC_SB);
    w.m1 = w.m0 + res_cap;
    w.run = w.m1 + res_cap;
    w.scores = w.run + (size_t)LOC_SB * 2 * pos_cap;
    w.pool = w.scores + LOC_SB * 10;
    w.stage_val = w.pool + pool_cap;
    w.stage_tag = (uint32_t *)(w.stage_val + 128);
    w.t_lo = w.stage_tag + (lean_tables ? 128 : 32);          /* (the hash route keeps one-byte tags) */
    w.t_off = w.t_lo + (lean_tables ? 32 : 0);
    w.tot = w.t_off + (lean_tables ? 33 : 0);
    w.c_idx = w.tot + LOC_SB * 2;
    w.c_pre = w.c_idx + LOC_SB;
    w.c_depth = (int32_t *)(w.c_pre + LOC_SB);
    w.c_cnt = (uint32_t *)(w.c_depth + LOC_SB);
    w.c_tr = w.c_cnt + LOC_SB * 2;
    w.pmk = (uint16_t *)(w.c_tr + LOC_SB * 2);
    w.cpre = w.pmk + (nl_tables ? (size_t)LOC_SB * 2 * pos_cap : 0);
    w.nlp = (uint8_t *)(w.cpre + (nl_tables ? (size_t)LOC_SB * 2 * pos_cap : 0));
    w.keep = w.nlp + 64;
    return w;
}

/* p0 >= 0: the retained table's offset and count are known already (the packed descriptor; ret_n fetched beside it) */
DEV void stage_tables(const BatchDev &b, const DevConfig *cfg, const K3Lds &k, uint32_t psm,
                      PeakTable *tab, NlTables *nl, bool with_table = true, int64_t p0 = -1, int R = 0) {
    const int lane = lane_id();
    if (with_table) {
        if (p0 >= 0) stage_peak_table_at(b, p0, R, k.t_e, tab);
        else stage_peak_table(b, psm, k.t_e, tab);
    } else if (p0 >= 0) {
        global_peak_table_at(b, psm, p0, R, tab);
    } else {
        global_peak_table(b, psm, tab);
    }
    nl->n_nl = cfg->n_nl;
    if (nl->n_nl) {
        for (int i = lane; i < (int)k.nl_cap; i += 64) k.nl_present[i] = cfg->present[i];
        if (lane < PYA_MAX_UNIQ) k.nl_uniq[lane] = cfg->uniq[lane];
    }
    nl->present = k.nl_present;
    nl->uniq = k.nl_uniq;
    wave_lds_sync();
    if (with_table) {
        grid_build(tab, k.grid);
        wave_lds_sync();
    }
}

/* ---------------------------------------------------------------------------------------
 * Batched localisation: the winner and up to sb-1 competitors at a time.
 *   1. one lane per (signature, direction) walks the residues and tabulates, per prefix
 *      length, the float32 running sum (and, with neutral losses, the loss sums that exist);
 *   2. the 10 depth scores of every signature of the batch are read off the score table from
 *      the cumulative counts score_signatures recorded;
 *   3. per (competitor, ion type) task the two fragment lists are written to an LDS pool and
 *      sorted if they can be out of order; every ion looks for partners within mz_error in the
 *      other list (its neighbours for position-indexed lists, a binary search otherwise), ions
 *      without a partner are staged and looked up a wave at a time; a task with a doubly
 *      partnered ion is replayed with the reference's serial greedy walk.
 * ------------------------------------------------------------------------------------- */
struct LocCtx {
    const BatchDev *b;
    const DevConfig *cfg;
    NlTables nl;
    PeakTable tab;
    LocLds w;
    int L, zmax;
    uint32_t pos_cap, pool_cap;
    int sb;                   /* signatures per batch (winner included), <= PYA_LOC_SB_MAX */
    int gtp;                  /* log2 of the ion types localised per pass                  */
    bool presorted;           /* charge 1, no neutral losses, all residue masses positive:  */
                              /* every fragment list comes out ascending, no check needed   */
    bool wide;                /* ... and every residue heavier than two tolerances: an ion has at most one
                               * partner, and only fragments between the first and the last residue in which
                               * two signatures differ can be site-determining (fused_core.hip.h) */
};

template <bool PLAIN>
DEV void loc_prefix_tables(const LocCtx &c, int S) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    if (lane < 2 * S) {
        const int s = lane >> 1, d = lane & 1;
        const uint64_t mask = w.sig_mask[s];
        const size_t base = (size_t)(s * 2 + d) * c.pos_cap;
        if (PLAIN || c.nl.n_nl == 0) {
            /* no neutral losses: one ion per prefix, so only the running sums are tabulated
             * (loc_site_ions knows pmk = 1 and cpre = step without reading them) */
            const uint64_t tmask = d ? (__brevll(mask) >> (64 - c.L)) : mask;   /* travel order */
            const uint32_t tlo = (uint32_t)tmask, thi = (uint32_t)(tmask >> 32);
            const float *p0 = w.m0 + (d ? c.L - 1 : 0), *p1 = w.m1 + (d ? c.L - 1 : 0);
            const int stride = d ? -1 : 1;
            float *dst = w.run + base;
            float running = 0.f;
            for (int step = 0; step + 1 < c.L; step++, p0 += stride, p1 += stride) {
                const uint32_t word = step < 32 ? tlo : thi;
                const bool mod = (word >> (step & 31)) & 1u;
                const float a = *p0, bmass = *p1;
                const float r = mod ? bmass : a;
                running = step == 0 ? r : r + running;
                dst[step] = running;
            }
            w.tot[s * 2 + d] = (uint32_t)(c.L - 1);
        } else if (!PLAIN) {
            float running = 0.f;
            uint32_t st = 0, cnt = 0;
            for (int step = 0; step + 1 < c.L; step++) {
                const int ri = d == 0 ? step : c.L - 1 - step;
                const bool mod = (mask >> ri) & 1ull;
                const float r = mod ? w.m1[ri] : w.m0[ri];
                running = step == 0 ? r : r + running;
                const uint32_t nlp = w.nlp[ri];
                const uint32_t cls = mod ? (nlp >> 4) : (nlp & 15u);
                if (cls) st = nl_bump(st, cls);
                const uint32_t pm = (uint32_t)c.nl.present[st & 255u];
                w.run[base + step] = running;
                w.pmk[base + step] = (uint16_t)pm;
                w.cpre[base + step] = (uint16_t)cnt;
                cnt += __popc(pm);
            }
            w.tot[s * 2 + d] = cnt;
        }
    }
}

DEV int ilog2_ceil(int v) {                     /* smallest g with (1 << g) >= v, v >= 1 */
    int g = 0;
    while ((1 << g) < v) g++;
    return g;
}

#define LOC_STAGE 128          /* surviving ions collected before they are looked up together */

/* Surviving ions are few and scattered: they are collected into a dense staging buffer and looked
 * up a full wave at a time.  `kept` lanes append (val, tag = competitor*2 + side); once 64 are
 * staged -- or at the last call, whatever is there -- one wave-wide lookup updates the
 * competitor's trial / match counts. */
DEV void loc_stage_flush(const LocCtx &c, int &staged) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    wave_lds_sync();
    const int take = staged < 64 ? staged : 64;
    if (lane < take) {
        const uint32_t tg = w.stage_tag[lane];
        atomicAdd(&w.c_tr[tg], 1u);
        if (match_rank(c.tab, w.stage_val[lane]) <= w.c_depth[tg >> 1]) atomicAdd(&w.c_cnt[tg], 1u);
    }
    wave_lds_sync();
    /* move the overflow (at most 63 entries) to the front */
    const int rem = staged - take;
    float mv = 0.f;
    uint32_t mt = 0;
    if (lane < rem) {
        mv = w.stage_val[take + lane];
        mt = w.stage_tag[take + lane];
    }
    wave_lds_sync();
    if (lane < rem) {
        w.stage_val[lane] = mv;
        w.stage_tag[lane] = mt;
    }
    staged = rem;
    wave_lds_sync();
}
DEV void loc_stage_push(const LocCtx &c, bool kept, float val, uint32_t tag, int &staged) {
    const LocLds &w = c.w;
    const uint64_t km = __ballot(kept);
    if (kept) {
        const int slot = staged + mask_rank(km);
        w.stage_val[slot] = val;
        w.stage_tag[slot] = tag;
    }
    staged += __popcll(km);
    if (staged >= 64) loc_stage_flush(c, staged);
}

/* Site-determining ions of competitors 1..S-1 against the winner (entry 0): fills w.c_cnt /
 * w.c_tr and the depth in w.c_depth.
 *
 * Fragment lists live in an LDS pool addressed [signature slot][type slot][P2] (type slots and P2
 * powers of two; slot 0 = the winner, generated once and shared by every task).  A "task" is one
 * (competitor, ion type): list A = winner, list B = competitor; its keep flags are
 * [task][side][P2]. */
/* PLAIN: returns true when the PSM needs a route only the general instantiation has (an ion with
 * two partners within mz_error: the reference's serial walk has to be replayed). */
template <bool PLAIN>
DEV bool loc_site_ions(const LocCtx &c, int S) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const DevConfig *cfg = c.cfg;
    const int T = cfg->n_types, Lm1 = c.L - 1;
    const int gt = c.gtp;                         /* ion types handled per pass = 1 << gt (power of two) */
    const uint64_t types64 = load_types64(cfg);
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
    /* longest list decides the (power of two) stride of the pool */
    uint32_t mmax = 1;
    if (lane < 2 * S) mmax = w.tot[lane] * (uint32_t)c.zmax;
    mmax = wave_max_u32(mmax);
    if (mmax < 1) mmax = 1;
    const int g2 = ilog2_ceil((int)mmax);
    const int P2 = 1 << g2;
    int per_round = (int)(c.pool_cap >> (gt + g2)) - 1;          /* competitor slots besides the winner */
    const int tpp = 1 << gt;
    if (per_round < 1) per_round = 1;
    const float err = cfg->mz_error;
    const FastDiv divL = fastdiv_make((uint32_t)(Lm1 > 0 ? Lm1 : 1));
    const FastDiv divM = fastdiv_make(mmax);
    wave_lds_sync();
    STAMP_BEGIN();
    /* ion types are taken tpp at a time so that the fragment lists of a pass fit a small pool */
    for (int tb = 0; tb < T; tb += tpp)
    for (int c0 = 1; c0 < S; c0 += per_round) {
        const int c1 = c0 + per_round < S ? c0 + per_round : S;
        const int ncomp = c1 - c0;
        /* signature slots of this round: slot 0 = winner (kept from the first round), slot j = c0+j-1 */
        const int slot_lo = c0 == 1 ? 0 : 1;
        const int nsl = 1 + ncomp;
        /* ---- generate: items (slot, type, prefix) ---- */
        const int gen_items = ((nsl - slot_lo) << gt) * Lm1;
        for (int e = lane; e < gen_items; e += 64) {
            const uint32_t li = fastdiv((uint32_t)e, divL);          /* (slot - slot_lo, type slot) */
            const int pos = e - (int)li * Lm1;
            const int t = (int)li & ((1 << gt) - 1), slot = slot_lo + ((int)li >> gt);
            if (tb + t >= T) continue;
            const int s = slot == 0 ? 0 : c0 + slot - 1;
            double A, B;
            type_constants(type_at(types64, tb + t), &A, &B);
            const int d = tb + t < cfg->n_fwd ? 0 : 1;
            const size_t idx = (size_t)(s * 2 + d) * c.pos_cap + pos;
            const float running = w.run[idx];
            if (PLAIN || c.nl.n_nl == 0) {
                float *dst = w.pool + ((size_t)((slot << gt) + t) << g2) + (size_t)pos * c.zmax;
                const double m = ((double)running + A) - B;
                for (int z = 1; z <= c.zmax; z++) *dst++ = charge_mz(m, z);
            } else if (!PLAIN) {
                uint32_t pm = w.pmk[idx];
                float *dst = w.pool + ((size_t)((slot << gt) + t) << g2) + (size_t)w.cpre[idx] * c.zmax;
                while (pm) {
                    const int v = __builtin_ctz(pm);
                    pm &= pm - 1;
                    const float x = running - c.nl.uniq[v];
                    const double m = ((double)x + A) - B;
                    for (int z = 1; z <= c.zmax; z++) *dst++ = charge_mz(m, z);
                }
            }
        }
        wave_lds_sync();
        STAMP_T(*c.b, 30, false);
        if (!PLAIN) {
        /* ---- out of order anywhere?  then pad to the stride and run the bitonic network ---- */
        const int nlists = nsl << gt;
        const int dense = nlists * (int)mmax;
        int unsorted = 0;
        for (int e = lane; !c.presorted && e < dense; e += 64) {
            const uint32_t lid = fastdiv((uint32_t)e, divM);
            const int i = e - (int)lid * (int)mmax;
            const int t = (int)lid & ((1 << gt) - 1), slot = (int)lid >> gt;
            if (tb + t >= T) continue;
            const int s = slot == 0 ? 0 : c0 + slot - 1;
            const int M = (int)w.tot[s * 2 + (tb + t < cfg->n_fwd ? 0 : 1)] * c.zmax;
            const float *base = w.pool + ((size_t)lid << g2);
            if (i + 1 < M && base[i] > base[i + 1]) unsorted = 1;
        }
        STAMP_T(*c.b, 31, false);
        if (__any(unsorted)) {
            for (int e = lane; e < (nlists << g2); e += 64) {
                const int lid = e >> g2, i = e & (P2 - 1);
                const int t = lid & ((1 << gt) - 1), slot = lid >> gt;
                int M = 0;
                if (tb + t < T) M = (int)w.tot[(slot == 0 ? 0 : c0 + slot - 1) * 2 + (tb + t < cfg->n_fwd ? 0 : 1)] * c.zmax;
                if (i >= M) w.pool[e] = __builtin_huge_valf();
            }
            wave_lds_sync();
            const int gh = g2 - 1;                               /* pairs per list = 1 << gh */
            for (int k = 2; k <= P2; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    /* the lists sit back to back at the stride P2 = 2 * (pairs per list), so a pair's index
                     * across all lists with a zero bit inserted at j is its lower element's pool index */
                    for (int e = lane; e < (nlists << gh); e += 64) {
                        const int lo = ((e & ~(j - 1)) << 1) | (e & (j - 1));
                        const bool up = (lo & k & (P2 - 1)) == 0;    /* (k = P2: the last merge, ascending in every list) */
                        float *at = w.pool + lo;
                        const float a = at[0], bb = at[j];
                        if ((a > bb) == up) {
                            at[0] = bb;
                            at[j] = a;
                        }
                    }
                    wave_lds_sync();
                }
            }
        }
        }
        STAMP_T(*c.b, 32, false);
        /* ---- cancel.  Site-determining ions = what the reference's greedy two-pointer walk
         * over the two sorted lists leaves (ModifiedPeptide.cpp:291-316).  When every ion has at
         * most one partner within mz_error in the other list, the walk cancels exactly those
         * pairs and emits everything else (an unpaired ion is strictly below/above every ion it
         * meets, because float subtraction is monotone) -- so pairs are found in parallel with
         * one binary search per ion.  A task in which some ion has two partners is replayed with
         * the serial walk, one task per lane.  Items: (task, side, index), dense. ---- */
        const int ntask = ncomp << gt;
        const int pair_items = ntask * 2 * (int)mmax;
        uint64_t bad_tasks = 0;
        /* Lists that are not position-indexed (several charges / neutral-loss variants merged by the
         * sort): most tasks have an ion with two partners, and replaying a whole task serially costs
         * 300 steps on one lane.  The greedy walk forgets everything at a gap: an ion that lies
         * mz_error or more above every ion before it (of both lists, in merged order, list A first
         * on ties) is reached with both cursors exactly past those ions, whatever happened before.
         * So every such ion starts its own walk -- one lane per cluster of ions chained by gaps below
         * mz_error, typically one to four steps -- and walks until the next ion would be such a start.
         * One binary search per ion (its predecessor in the other list) finds the starts. */
        const bool clusters = !PLAIN && !c.presorted && !(c.b->debug & 4096);
        if (clusters) {
            for (int base = 0; base < pair_items; base += 64) {
                const int e = base + lane;
                if (e >= pair_items) continue;
                const uint32_t ts = fastdiv((uint32_t)e, divM);     /* task*2 + side */
                const int i = e - (int)ts * (int)mmax;
                const int side = (int)ts & 1, task = (int)ts >> 1;
                const int t = task & ((1 << gt) - 1), cj = task >> gt;
                if (tb + t >= T) continue;
                const int cc = c0 + cj;
                const int d = tb + t < cfg->n_fwd ? 0 : 1;
                const int na = (int)w.tot[0 * 2 + d] * c.zmax, nb = (int)w.tot[cc * 2 + d] * c.zmax;
                if (i >= (side ? nb : na)) continue;
                const float *la = w.pool + ((size_t)t << g2);
                const float *lb = w.pool + ((size_t)(((cj + 1) << gt) + t) << g2);
                const float *mine = side ? lb : la, *other = side ? la : lb;
                const int Mo = side ? na : nb;
                const float me = mine[i];
                /* ions of the other list that come before this one in merged order */
                int j = 0;                                   /* in 0 .. Mo (a list can fill its stride: start at P2) */
                for (int step = P2; step > 0; step >>= 1) {
                    const int probe = j + step;
                    const float o = probe - 1 < Mo ? other[probe - 1] : __builtin_huge_valf();
                    if (side ? (o <= me) : (o < me)) j = probe;
                }
                const float before_own = i > 0 ? mine[i - 1] : -__builtin_huge_valf();
                const float before_other = j > 0 ? other[j - 1] : -__builtin_huge_valf();
                if (!(me - before_own >= err && me - before_other >= err)) continue;
                uint8_t *ka = w.keep + ((size_t)(task * 2) << g2);
                uint8_t *kb = ka + P2;
                int ia = side ? j : i, ib = side ? i : j;
                float reached = -__builtin_huge_valf();      /* largest ion this walk has consumed */
                for (bool first = true;; first = false) {
                    const float x = ia < na ? la[ia] : __builtin_huge_valf();
                    const float y = ib < nb ? lb[ib] : __builtin_huge_valf();
                    const float next = x <= y ? x : y;
                    if (next == __builtin_huge_valf()) break;                 /* both lists done */
                    if (!first && next - reached >= err) break;              /* the next walk's start */
                    if (__builtin_fabsf(x - y) < err) {          /* ModifiedPeptide.cpp:291-316 */
                        ka[ia++] = 0;
                        kb[ib++] = 0;
                        reached = __builtin_fmaxf(reached, x > y ? x : y);
                    } else if (x < y) {
                        ka[ia++] = 1;
                        reached = __builtin_fmaxf(reached, x);
                    } else {
                        kb[ib++] = 1;
                        reached = __builtin_fmaxf(reached, y);
                    }
                }
            }
            wave_lds_sync();
            STAMP_T(*c.b, 33, false);
            STAMP_T(*c.b, 37, false);
        } else {

        /* Optimistic: an ion without a partner is staged for its lookup right here.  Should a task
         * turn out to need the serial replay (rare), the counts are put back and the round is
         * redone from the keep flags. */
        uint32_t snap_tr = 0, snap_cnt = 0;
        if (lane < S * 2) {
            snap_tr = w.c_tr[lane];
            snap_cnt = w.c_cnt[lane];
        }
        int staged = 0;
        /* Lean route with `wide` residues: only the steps between the first and the last residue in which
         * the winner and the competitor differ are looked at (one ion per step here); the items of a task
         * are the two sides of its span, tasks back to back. */
        const bool spans = PLAIN && c.wide && ntask <= 32;
        int n_items = pair_items;
        if (spans) {
            if (lane < ntask) {
                const int t = lane & ((1 << gt) - 1), cc = c0 + (lane >> gt);
                uint32_t lo = 0, len = 0;
                if (tb + t < T) {
                    const uint64_t diff = w.sig_mask[0] ^ w.sig_mask[cc];         /* residue positions */
                    const int r_lo = __builtin_ctzll(diff), r_hi = 63 - __builtin_clzll(diff);
                    const int d = tb + t < cfg->n_fwd ? 0 : 1;
                    lo = (uint32_t)(d ? c.L - 1 - r_hi : r_lo);
                    len = (uint32_t)(r_hi - r_lo);
                }
                w.t_lo[lane] = lo;
                w.t_off[lane] = 2u * len;                      /* both sides */
            }
            wave_lds_sync();
            uint32_t acc = 0;
            for (int tk = 0; tk < ntask; tk++) {              /* (every lane, same values) */
                const uint32_t n = w.t_off[tk];
                wave_lds_sync();
                if (lane == 0) w.t_off[tk] = acc;
                acc += n;
            }
            if (lane == 0) w.t_off[ntask] = acc;
            n_items = (int)acc;
            wave_lds_sync();
        }
        for (int base = 0; base < n_items; base += 64) {         /* wave-uniform trip count */
            const int e = base + lane;
            bool multi = false, kept = false;
            float me = 0.f;
            uint32_t tag = 0;
            if (e < n_items) {
                uint32_t ts;                                     /* task*2 + side */
                int i;
                if (spans) {
                    int task = 0;
                    for (int tk = 1; tk < ntask; tk++) task += (uint32_t)e >= w.t_off[tk] ? 1 : 0;
                    const int rem = e - (int)w.t_off[task];
                    const int len = (int)(w.t_off[task + 1] - w.t_off[task]) >> 1;
                    const int sd = rem >= len ? 1 : 0;
                    ts = (uint32_t)(task * 2 + sd);
                    i = (int)w.t_lo[task] + rem - sd * len;
                } else {
                    ts = fastdiv((uint32_t)e, divM);
                    i = e - (int)ts * (int)mmax;
                }
                const int side = (int)ts & 1, task = (int)ts >> 1;
                const int t = task & ((1 << gt) - 1), cj = task >> gt;   /* competitor slot - 1 */
                if (tb + t < T) {
                    const int cc = c0 + cj;
                    const int d = tb + t < cfg->n_fwd ? 0 : 1;
                    const int M = (int)w.tot[(side ? cc : 0) * 2 + d] * c.zmax;
                    const int Mo = (int)w.tot[(side ? 0 : cc) * 2 + d] * c.zmax;
                    if (i < M) {
                        const float *mine = w.pool + ((size_t)(((side ? cj + 1 : 0) << gt) + t) << g2);
                        const float *other = w.pool + ((size_t)(((side ? 0 : cj + 1) << gt) + t) << g2);
                        me = mine[i];
                        /* diff is always (list A) - (list B), as the reference computes it.  Seen
                         * from an A ion the B list ascends, so diff descends: skip B ions with
                         * diff >= err.  Seen from a B ion diff ascends: skip A ions with diff <= -err. */
                        int cnt = -1;
                        if (PLAIN || c.presorted) {
                            /* position-indexed ascending lists: the first candidate partner sits at
                             * the ion's own index or one above.  Four neighbours fetched together
                             * decide it without the dependent probes of a binary search: index q is
                             * the search result iff other[q-1] is skipped and other[q] is not
                             * (the skip test is monotone along an ascending list). */
                            float d[4];
                            bool ok[4], sk[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const int q = i - 1 + u;
                                ok[u] = q >= 0 && q < Mo;
                                const float o = ok[u] ? other[q] : (q < 0 ? -__builtin_huge_valf() : __builtin_huge_valf());
                                d[u] = side ? (o - me) : (me - o);
                                sk[u] = side ? (d[u] <= -err) : (d[u] >= err);
                            }
                            const int w1 = (ok[1] && __builtin_fabsf(d[1]) < err) ? 1 : 0;
                            const int w2 = (ok[2] && __builtin_fabsf(d[2]) < err) ? 1 : 0;
                            const int w3 = (ok[3] && __builtin_fabsf(d[3]) < err) ? 1 : 0;
                            if (sk[0] && !sk[1]) cnt = w1 + w2;            /* search result = i     */
                            else if (sk[1] && !sk[2]) cnt = w2 + w3;       /* search result = i + 1 */
                        }
                        if (cnt < 0) {
                            int j = 0;
                            for (int step = P2 >> 1; step > 0; step >>= 1) {
                                const int probe = j + step;
                                const float o = probe - 1 < Mo ? other[probe - 1] : __builtin_huge_valf();
                                const float diff = side ? (o - me) : (me - o);
                                const bool skip = side ? (diff <= -err) : (diff >= err);
                                if (skip) j = probe;
                            }
                            cnt = 0;
                            for (int q = j; q < j + 2 && q < Mo; q++) {
                                const float o = other[q];
                                const float diff = side ? (o - me) : (me - o);
                                cnt += (__builtin_fabsf(diff) < err) ? 1 : 0;
                            }
                        }
                        kept = cnt == 0;
                        w.keep[((size_t)ts << g2) + i] = kept ? 1 : 0;
                        multi = cnt > 1;
                        tag = (uint32_t)(cc * 2 + side);
                    }
                }
            }
            if (PLAIN && __any(multi)) return true;
            uint64_t rest = PLAIN ? 0ull : __ballot(multi);
            while (rest) {
                const int src = __builtin_ctzll(rest);
                rest &= rest - 1;
                bad_tasks |= 1ull << (fastdiv((uint32_t)(base + src), divM) >> 1);
            }
            if (!bad_tasks) loc_stage_push(c, kept, me, tag, staged);
        }
        if (!bad_tasks) {
            while (staged > 0) loc_stage_flush(c, staged);
            wave_lds_sync();
            STAMP_T(*c.b, 33, false);
            continue;
        }
        if (PLAIN) continue;                               /* (unreachable: bad_tasks is empty) */
        /* ---- a task needs the reference's serial walk: undo the optimistic counts ---- */
        wave_lds_sync();
        if (lane < S * 2) {
            w.c_tr[lane] = snap_tr;
            w.c_cnt[lane] = snap_cnt;
        }
        wave_lds_sync();
        STAMP_T(*c.b, 33, false);
        {
            const int task = lane;                               /* ntask <= 64 */
            if (task < ntask && ((bad_tasks >> task) & 1ull)) {
                const int t = task & ((1 << gt) - 1), cj = task >> gt;
                const int cc = c0 + cj;
                const int d = tb + t < cfg->n_fwd ? 0 : 1;
                const int na = (int)w.tot[0 * 2 + d] * c.zmax, nb = (int)w.tot[cc * 2 + d] * c.zmax;
                const float *la = w.pool + ((size_t)t << g2);
                const float *lb = w.pool + ((size_t)(((cj + 1) << gt) + t) << g2);
                uint8_t *ka = w.keep + ((size_t)(task * 2) << g2);
                uint8_t *kb = ka + P2;
                int i = 0, j = 0;
                while (i < na || j < nb) {
                    if (j == nb) {
                        ka[i++] = 1;
                    } else if (i == na) {
                        kb[j++] = 1;
                    } else {
                        const float x = la[i], y = lb[j];
                        if (__builtin_fabsf(x - y) < err) {
                            ka[i++] = 0;
                            kb[j++] = 0;
                        } else if (x < y) {
                            ka[i++] = 1;
                        } else {
                            kb[j++] = 1;
                        }
                    }
                }
            }
            wave_lds_sync();
        }
        STAMP_T(*c.b, 37, false);
        }
        /* ---- match the surviving ions from the keep flags ---- */
        int staged = 0;
        for (int base = 0; base < pair_items; base += 64) {
            const int e = base + lane;
            bool kept = false;
            float val = 0.f;
            uint32_t tag = 0;
            if (e < pair_items) {
                const uint32_t ts = fastdiv((uint32_t)e, divM);
                const int i = e - (int)ts * (int)mmax;
                const int side = (int)ts & 1, task = (int)ts >> 1;
                const int t = task & ((1 << gt) - 1), cj = task >> gt;
                if (tb + t < T) {
                    const int cc = c0 + cj;
                    const int M = (int)w.tot[(side ? cc : 0) * 2 + (tb + t < cfg->n_fwd ? 0 : 1)] * c.zmax;
                    if (i < M && w.keep[((size_t)ts << g2) + i]) {
                        kept = true;
                        val = w.pool[((size_t)(((side ? cj + 1 : 0) << gt) + t) << g2) + i];
                        tag = (uint32_t)(cc * 2 + side);
                    }
                }
            }
            loc_stage_push(c, kept, val, tag, staged);
        }
        while (staged > 0) loc_stage_flush(c, staged);
        wave_lds_sync();
        STAMP_T(*c.b, 34, false);
    }
    return false;
}

#include "localize_hash.hip.h"

/* Ascores of every modified site of the winner (cpp/Ascore.cpp:212-254): walks the pushed
 * competitors sb-1 at a time.  `rec` holds the cumulative counts score_signatures wrote, from which
 * the depth scores are read off the score table (the same reads score_signatures made).  Lane a
 * accumulates site a in *my_asc; alternative sites go to site_alt[a] (LDS). */
/* The cumulative rank counts of signatures [from, S) of the batch, counted again instead of read from the
 * records score_signatures writes: every (signature, direction, step) fragment of the batch's prefix tables is
 * looked up in the peak table (LDS) and tallied.  Same running sums, same m/z arithmetic, same window test as
 * the walkers, so the same counts; used where the records never went to memory (score_big's own localisation:
 * thousands of site assignments, a handful of them ever looked at).  Plain settings: charge 1, one ion type
 * per direction, no neutral losses.  rec_batch: [sb][PYA_REC_WORDS] in the records' layout; hist: [sb][PYA_NTOP]. */
DEV void loc_recount(const LocCtx &c, int from, int S, uint32_t *rec_batch, uint32_t *hist) {
    const int lane = lane_id();
    const LocLds &w = c.w;
    const DevConfig *cfg = c.cfg;
    const int Lm1 = c.L - 1;
    const int n_f = cfg->n_fwd, n_b = cfg->n_types - cfg->n_fwd;
    const int ndir = (n_f > 0 ? 1 : 0) + (n_b > 0 ? 1 : 0);
    for (int i = lane; i < S * PYA_NTOP; i += 64) hist[i] = 0u;
    wave_lds_sync();
    double Af = 0., Bf = 0., Ab = 0., Bb = 0.;
    if (n_f > 0) type_constants(cfg->types[0], &Af, &Bf);
    if (n_b > 0) type_constants(cfg->types[n_f], &Ab, &Bb);
    const FastDiv divL = fastdiv_make((uint32_t)(Lm1 > 0 ? Lm1 : 1));
    const int items = (S - from) * 2 * Lm1;
    for (int base = 0; base < items; base += 64) {
        const int e = base + lane;
        if (e < items) {
            const uint32_t sd = fastdiv((uint32_t)e, divL);
            const int step = e - (int)sd * Lm1;
            const int s = from + (int)(sd >> 1), d = (int)(sd & 1u);
            if (d == 0 ? n_f > 0 : n_b > 0) {
                const float running = w.run[(size_t)(s * 2 + d) * c.pos_cap + step];
                const double m = ((double)running + (d ? Ab : Af)) - (d ? Bb : Bf);
                const int rk = match_rank(c.tab, (float)(m + 1.007825));
                if (rk < PYA_NTOP) atomicAdd(&hist[s * PYA_NTOP + rk], 1u);
            }
        }
    }
    wave_lds_sync();
    if (lane >= from && lane < S) {
        uint32_t cum[PYA_NTOP], acc = 0;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d++) {
            acc += hist[lane * PYA_NTOP + d];
            cum[d] = acc;
        }
        uint32_t *r6 = rec_batch + (size_t)lane * PYA_REC_WORDS;
#pragma unroll
        for (int d = 0; d < PYA_NTOP; d += 2) r6[d >> 1] = cum[d] | (cum[d + 1] << 16);
        r6[5] = (uint32_t)(ndir * Lm1);
    }
    wave_lds_sync();
}

/* rec_batch / hist (LDS, optional): the counts of the batch's signatures are recounted (loc_recount) instead
 * of read from `rec` */
/* HASH: site-determining ions by loc_site_ions_hash over *hl (it can decline: returns true, nothing written) */
template <bool PLAIN, bool HASH = false>
DEV bool loc_ascore_all(LocCtx &ctx, const PushedEntry *pushed, uint32_t np, unsigned long long *site_alt,
                        const uint32_t *rec, uint64_t best_bits, float best_ws,
                        uint32_t best_i, uint64_t site_mask, float *my_asc_io, uint64_t *my_alt_io,
                        int *fail_io, uint32_t *rec_batch = nullptr, uint32_t *hist = nullptr,
                        const HashLds *hl = nullptr) {
    const int lane = lane_id();
    const BatchDev &b = *ctx.b;
    const LocLds &w = ctx.w;
    float my_asc = *my_asc_io;
    uint64_t my_alt = *my_alt_io;
    int fail = *fail_io;
    bool have_best = false;
    uint32_t e = 0;
    STAMP_BEGIN();
    while (e < np) {
        /* the next sb-1 competitors, one per lane (the scan pushed no exact tie of the winner:
         * those were settled there, Ascore.cpp:159-161) */
        const int take = (int)(np - e) < ctx.sb - 1 ? (int)(np - e) : ctx.sb - 1;
        const int S = 1 + take;
        if (lane >= 1 && lane < S) {
            const PushedEntry pe = pushed[e + lane - 1];
            const uint64_t c = pe.bits;
            const uint64_t gone = best_bits & ~c, came = c & ~best_bits;
            const int a = __popcll(best_bits & (gone - 1));
            atomicOr(&site_alt[a], 1ull << nth_set_bit(site_mask, __builtin_ctzll(came)));
            w.sig_mask[lane] = deposit_sites(c, site_mask);
            w.c_idx[lane] = (uint32_t)a;
            w.c_pre[lane] = pe.idx;
        }
        e += (uint32_t)take;
        STAMP_T(b, 26, false);
        if (S == 1) continue;
        if (lane == 0) w.c_pre[0] = best_i;
        wave_lds_sync();
        if (!(b.debug & 4)) loc_prefix_tables<PLAIN>(ctx, S);
        wave_lds_sync();
        STAMP_T(b, 27, false);
        {
            /* depth scores of signatures [have_best, S) from the recorded cumulative counts */
            if (rec_batch) loc_recount(ctx, have_best ? 1 : 0, S, rec_batch, hist);
            for (int i = (have_best ? 10 : 0) + lane; i < S * 10; i += 64) {
                const int s = i / 10, d = i % 10;
                const uint32_t *r6 = rec_batch ? rec_batch + (size_t)s * PYA_REC_WORDS : rec + (size_t)w.c_pre[s] * PYA_REC_WORDS;
                const uint32_t cum = (r6[d >> 1] >> ((d & 1) * 16)) & 0xffffu;
                const uint32_t nf = r6[5];
                float sc = 0.f;
                if (nf <= b.lut_n_max) sc = b.lut[lut_row(nf) + (uint32_t)d * (nf + 1) + cum];
                else fail = 1;
                w.scores[i] = sc;
            }
            wave_lds_sync();
            STAMP_T(b, 28, false);
        }
        have_best = true;
        if (!(b.debug & 1)) {
            if (HASH) {
                if (loc_site_ions_hash(ctx, *hl, S)) return true;
            } else if (loc_site_ions<PLAIN>(ctx, S)) return true;
        }
        STAMP_T(b, 35, false);
        /* one competitor per lane: the table reads of all of them are in flight together */
        float asc_l = 0.f;
        if (lane >= 1 && lane < S) {
            const uint32_t tr0 = w.c_tr[lane * 2], tr1 = w.c_tr[lane * 2 + 1];
            const uint32_t n0 = w.c_cnt[lane * 2], n1 = w.c_cnt[lane * 2 + 1];
            const uint32_t depth = (uint32_t)w.c_depth[lane];
            if (tr0 > b.lut_n_max || tr1 > b.lut_n_max) {
                fail = 1;
            } else {
                const float sc0 = b.lut[lut_row(tr0) + depth * (tr0 + 1) + n0];
                const float sc1 = b.lut[lut_row(tr1) + depth * (tr1 + 1) + n1];
                asc_l = sc0 - sc1;
            }
        }
        for (int cc = 1; cc < S; cc++) {
            const float asc = __shfl(asc_l, cc);
            if (lane == (int)w.c_idx[cc]) my_asc = asc < my_asc ? asc : my_asc;
        }
        wave_lds_sync();
    }
    *my_asc_io = my_asc;
    *my_alt_io = my_alt;
    *fail_io = fail;
    return false;
}

#endif
