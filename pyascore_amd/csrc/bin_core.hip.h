/* bin_core.hip.h -- raw peaks of one spectrum -> retained-peak table in LDS (one wavefront).
 * See bin_spectra.hip for the notes. */
#ifndef PYA_BIN_CORE_H
#define PYA_BIN_CORE_H
#include "device_common.hip.h"


/* ---------------------------------------------------------------------------------------
 * Exact top-n_top selection of one window when intensities tie (Spectra.cpp:24-41): libstdc++'s
 * std::nth_element (introselect: median-of-3 partitions, heap select at depth 0, closing insertion
 * sort of <= 3) followed by resize and std::sort (<= 16 elements: a stable insertion sort), run
 * serially by one lane on an index array; keys are read through the indices.  comp(a, b) =
 * intensity[a] > intensity[b].
 * ------------------------------------------------------------------------------------- */
/* Two element stores:
 *   RunDirect   -- (intensity, index) pairs permuted in place (the window's slice of the LDS
 *                  intensity array plus a parallel index array): one LDS read per key;
 *   RunIndirect -- only an index list is permuted, keys are read through it (windows whose peaks
 *                  are scattered over the spectrum: unordered input). */
struct RunElem {
    double k;
    uint16_t i;
};
struct RunDirect {
    double *key;            /* [len] */
    uint16_t *idx;          /* [len] */
    DEV double k(int pos) const { return key[pos]; }
    DEV RunElem get(int pos) const { return {key[pos], idx[pos]}; }
    DEV void put(int pos, const RunElem &e) const {
        key[pos] = e.k;
        idx[pos] = e.i;
    }
    DEV uint16_t id(int pos) const { return idx[pos]; }
};
struct RunIndirect {
    uint16_t *idx;          /* [len] peak indices */
    const double *key;      /* the spectrum's intensities */
    DEV double k(int pos) const { return key[idx[pos]]; }
    DEV RunElem get(int pos) const {
        const uint16_t i = idx[pos];
        return {key[i], i};
    }
    DEV void put(int pos, const RunElem &e) const { idx[pos] = e.i; }
    DEV uint16_t id(int pos) const { return idx[pos]; }
};
template <class R>
DEV void run_swap(const R &r, int a, int b) {
    const RunElem x = r.get(a), y = r.get(b);
    r.put(a, y);
    r.put(b, x);
}
/* __adjust_heap + __push_heap on [first, first+len) */
template <class R>
DEV void run_heap_adjust(const R &r, int first, int hole, int len, const RunElem &v) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (r.k(first + child) > r.k(first + child - 1)) child--;
        r.put(first + hole, r.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        r.put(first + hole, r.get(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && r.k(first + parent) > v.k) {
        r.put(first + hole, r.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    r.put(first + hole, v);
}
/* __heap_select(first, middle, last) */
template <class R>
DEV void run_heap_select(const R &r, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; parent--) {
            run_heap_adjust(r, first, parent, len, r.get(first + parent));
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; i++) {
        if (r.k(i) > r.k(first)) {                            /* __pop_heap(first, middle, i) */
            const RunElem v = r.get(i);
            r.put(i, r.get(first));
            run_heap_adjust(r, first, 0, len, v);
        }
    }
}
/* __unguarded_partition_pivot(f, l) */
template <class R>
DEV int run_partition(const R &r, int f, int l) {
    const int mid = f + (l - f) / 2;
    const double a = r.k(f + 1), b = r.k(mid), c = r.k(l - 1);
    int pick;
    if (a > b) {
        if (b > c) pick = mid;
        else if (a > c) pick = l - 1;
        else pick = f + 1;
    } else if (a > c) pick = f + 1;
    else if (b > c) pick = l - 1;
    else pick = mid;
    run_swap(r, f, pick);
    const double pv = r.k(f);
    int lo = f + 1, hi = l;
    for (;;) {
        while (r.k(lo) > pv) lo++;
        hi--;
        while (pv > r.k(hi)) hi--;
        if (!(lo < hi)) return lo;
        run_swap(r, lo, hi);
        lo++;
    }
}
/* stable insertion sort of positions [first, last): what __insertion_sort leaves */
template <class R>
DEV void run_insertion_sort(const R &r, int first, int last) {
    for (int i = first + 1; i < last; i++) {
        const RunElem v = r.get(i);
        int j = i - 1;
        while (j >= first && v.k > r.k(j)) {
            r.put(j + 1, r.get(j));
            j--;
        }
        r.put(j + 1, v);
    }
}
/* Ranks of one window whose len elements sit in r in input order: rank_out[element id] = rank
 * inside the window, 255 for the ones that are not retained; ntop = peaks retained per window (DevConfig.n_top). */
template <class R>
DEV void run_exact_ranks(const R &r, int len, uint8_t *rank_out, int ntop) {
    for (int e = 0; e < len; e++) rank_out[r.id(e)] = 255;
    if (len > ntop) {                                          /* std::nth_element(begin, begin + n_top - 1, end) */
        int first = 0, last = len;
        const int nth = ntop - 1;
        int depth = 0;
        for (int t = len; t > 1; t >>= 1) depth++;
        depth *= 2;
        bool done = false;
        while (last - first > 3) {
            if (depth == 0) {
                run_heap_select(r, first, nth + 1, last);
                run_swap(r, first, nth);
                done = true;
                break;
            }
            depth--;
            const int cut = run_partition(r, first, last);
            if (cut <= nth) first = cut;
            else last = cut;
        }
        if (!done) run_insertion_sort(r, first, last);
    }
    const int n = len < ntop ? len : ntop;                     /* resize(n_top); std::sort */
    run_insertion_sort(r, 0, n);
    for (int q = 0; q < n; q++) rank_out[r.id(q)] = (uint8_t)q;
}

/* Bins the spectrum of `psm` (Spectra.cpp:43-68, :24-41).  On return *out_mz / *out_rank point
 * into `lds` and hold the retained peaks (ascending float32 m/z, rank inside their window);
 * returns their count, or -1 with *status set.  Ends with an LDS sync.
 *
 * Two bodies in two kernels, because the rare paths would otherwise double the registers of the common one:
 * bin_fast takes the common case -- peaks in m/z order, at most 64 windows, finite non-negative intensities, no
 * two equal intensities inside a window's top n_top -- and returns PYA_BIN_REDO for everything else; bin_exact
 * handles everything (any peak order, ties resolved as std::nth_element + std::sort do). */
#define PYA_BIN_REDO (-2)
#ifndef BIN_BLOCK
#define BIN_BLOCK 6
#endif
#define PYA_BIN_FAST_WINDOWS 64    /* window ids the composite key has room for */
#define PYA_BIN_KEY_BITS 25        /* intensity bits of the composite key: 5 exponent + 20 mantissa bits below the largest */

/* whole-wave DPP shifts (GFX9): lane i takes the value of lane i - 1 / i + 1; lane 0 / 63 take `edge`.  One
 * vector move where __shfl_up / __shfl_down are an LDS permute with its address arithmetic.
 * (scripts/dpp_probe.hip checks the semantics on the device.) */
DEV uint32_t lane_prev_u32(uint32_t x, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
DEV uint32_t lane_next_u32(uint32_t x, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
DEV double lane_next_f64(double x, double edge) {
    const uint64_t xb = (uint64_t)__double_as_longlong(x), eb = (uint64_t)__double_as_longlong(edge);
    const uint32_t lo = lane_next_u32((uint32_t)xb, (uint32_t)eb), hi = lane_next_u32((uint32_t)(xb >> 32), (uint32_t)(eb >> 32));
    return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
}
DEV double first_lane_f64(double x) {
    const uint64_t xb = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xb >> 32));
    return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
}

/* LDS of one wavefront of the binning stage: the general body's 15 bytes per peak (cap a multiple of 32) and its window
 * starts; the common-case body needs 13 bytes per peak and its window table */
#define PYA_BIN_WAVE_BYTES(cap) ((((size_t)(cap) * 15 + 63) & ~(size_t)63) + 320)
#define PYA_BIN_FAST_BYTES(cap) ((((size_t)(cap) * 13 + 63) & ~(size_t)63) + 320)

/* number of set bits of m below this lane */
DEV uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
/* next lane's value; lane 63 gets zeros (no copy of the old value, as lane_next_* need) */
DEV double lane_next_f64_or_zero(double x) {
    const uint64_t xb = (uint64_t)__double_as_longlong(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)xb, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(xb >> 32), 0x130, 0xf, 0xf, true);
    return __longlong_as_double((long long)((uint64_t)lo | ((uint64_t)hi << 32)));
}

/* The common case.  LDS (PYA_BIN_FAST_BYTES):
 *   ckey u32[2 cap]  one composite key per peak, (63 - window) << 25 | intensity key, then zeros for the longest window
 *   mzf  f32[cap]    float m/z per peak; the retained m/z are compacted into it in place
 *   rank u8 [cap]    ranks of the retained peaks (output)
 *   wtab u16[65 + 64] last peak of every window (and a slot for "the window before the first"), first peak of every window
 *   cmax u32[15]      spectra of up to 960 peaks: per chunk of 64 peaks, the longest window that reaches into it
 * A sorted spectrum's windows are runs of consecutive peaks, so a peak's rank is the number of run mates that are more
 * intense.  The mates are read from the run's first peak on, straight through its end, for as many steps as the
 * longest run of the spectrum has peaks -- the same trip count for every lane, no per-lane bounds: what follows a run
 * in LDS belongs to later windows (or is the zero padding), and the window field makes all of that compare below
 * every peak of this run.  The intensity key is the high word of the float64 intensity (sign 0, order-preserving)
 * minus a base that leaves room for 2^32 of dynamic range below the spectrum's most intense peak; "more intense" is
 * then the sign of a 32-bit difference: subtract, shift, add -- three instructions of the cheap class per mate
 * (profiles/r03_valu_ceiling.md) where a float64 compare with its masks took four of the expensive one.
 * Keys that are equal (intensities that agree in 20 mantissa bits, or lie 2^32 below the maximum) make the counts of a
 * window fall short of len (len - 1) / 2, which the deficit notices; only for a peak that ranks inside the top n_top
 * does it matter: those are then ordered by their whole intensities, and if these are equal too the spectrum is handed
 * over (std::nth_element decides between equal intensities).
 *
 * DIRECT: the retained peaks go straight to the workspace table as the sweep finds them (8 contiguous bytes per
 * retained peak and lane) instead of being compacted in LDS for bin_store -- the batch kernel's way; the one-PSM
 * kernels keep the LDS table for the stage that follows in the same wavefront.
 *
 * PRE: the caller has the first block of the spectrum in registers already (the one-PSM kernel issues these loads
 * before anything else: its spectrum sits in host memory, a round trip of microseconds) and knows the peak count;
 * the spectrum starts at offset 0 of b.mz / b.inten then. */
struct BinPre {
    uint32_t P;                 /* peaks */
    double v[BIN_BLOCK];        /* m/z of peak u * 64 + lane (past the end: the last peak's) */
    uint32_t hw[BIN_BLOCK];     /* high word of its intensity */
    double mx;                  /* m/z of the last peak */
};
/* the loads behind a BinPre (all issued, none waited for) */
DEV void bin_preload(const double *mz, const double *inten, uint32_t P, BinPre *pre) {
    const uint32_t *inten_hi = (const uint32_t *)inten + 1;
    pre->P = P;
#pragma unroll
    for (uint32_t u = 0; u < BIN_BLOCK; u++) {
        const uint32_t i = u * 64 + (uint32_t)lane_id();
        const uint32_t ic = i < P ? i : P - 1;
        pre->v[u] = mz[ic];
        pre->hw[u] = inten_hi[2 * ic];
    }
    pre->mx = mz[P - 1];
}

template <bool DIRECT, bool PRE = false>
DEV int bin_fast(const BatchDev &b, uint32_t psm, unsigned char *lds, uint32_t cap, const float **out_mz,
                 const uint8_t **out_rank, int *status, const BinPre *pre = nullptr) {
    const int lane = lane_id();
    uint32_t *ckey = (uint32_t *)lds;
    float *s_mzf = (float *)(ckey + 2 * (size_t)cap);
    uint8_t *o_rank = (uint8_t *)(s_mzf + cap);
    uint16_t *w_last = (uint16_t *)(lds + (((size_t)cap * 13 + 63) & ~(size_t)63));   /* [64] + the slot of "window 64" */
    uint16_t *w_first = w_last + PYA_BIN_FAST_WINDOWS + 1;

    STAMP_BEGIN();
    STAMP_T(b, 1, -1);
    const int64_t p0 = PRE ? 0 : b.peak_off[psm];
    const uint32_t P = PRE ? pre->P : (uint32_t)(b.peak_off[psm + 1] - p0);
    const double *mz = b.mz + p0;
    const double *inten = b.inten + p0;
    const float bin_size = b.cfg->bin_size;
    const int ntop = b.cfg->n_top;                           /* peaks retained per window */
    /* (an estimate of the reciprocal is enough: the quotients below are corrected with an exact remainder) */
    const double bsd = (double)bin_size, inv_bs = __builtin_amdgcn_rcp(bsd);
    /* a sorted spectrum has its extremes at the ends */
    const double mn = PRE ? first_lane_f64(pre->v[0]) : mz[0], mx = PRE ? first_lane_f64(pre->mx) : mz[P - 1];
    *status = PYA_ST_OK;
    /* window bounds from the extremes (Spectra.cpp:46-48): floor(mn / 100.) and ceil(mx / 100.) as the reference's
     * double divisions give them, without the divisions -- see window_of below: 100 k is exact in double for every
     * integer k in reach, so the rounded quotient reaches k exactly when the true one does, and floor / ceil of the
     * true quotient follow from an estimate and the exact remainder. */
    auto div100 = [](double v, bool up) -> double {
        double q = __builtin_floor(v * 0.01);
        double r = __builtin_fma(-q, 100., v);
        if (r >= 100.) {
            q += 1.;
            r -= 100.;
        } else if (r < 0.) {
            q -= 1.;
            r += 100.;
        }
        return up && r > 0. ? q + 1. : q;
    };
    const float min_mz = (float)(div100(mn, false) * 100.);
    const float max_mz = (float)(div100(mx, true) * 100.);
    const float nb_f = __builtin_ceilf((max_mz - min_mz) / bin_size);        /* float arithmetic, :48 */
    const bool ok = nb_f >= 1.f && nb_f <= 65535.f;
    const uint32_t n_bins = ok ? (uint32_t)nb_f : 1u;
    /* (more windows than the keys and the window table have room for: handed over below; until then the ids stay
     * inside the table) */
    const int last_win = (int)(n_bins < PYA_BIN_FAST_WINDOWS ? n_bins : PYA_BIN_FAST_WINDOWS) - 1;
    /* window id of one peak (double arithmetic, Spectra.cpp:55-58): floor((v - min) / bin_size) as the
     * reference's double division gives it, without the division: every multiple k * bin_size
     * (k < 2^16, bin_size a float) is exact in double, so the rounded quotient reaches k exactly when
     * the true one does and the floor equals the mathematical one -- which a reciprocal estimate plus
     * an exact remainder (fma) pins down.  (A NaN or a negative quotient converts to a window below 0: clamped too.) */
    auto window_of = [&](double v) -> uint32_t {
        const double x = v - (double)min_mz;
        const double q = __builtin_floor(x * inv_bs);
        const double r = __builtin_fma(-q, bsd, x);
        int qi = (int)q;
        qi += r >= bsd ? 1 : 0;
        qi -= r < 0. ? 1 : 0;
        int w;                                               /* (the clamp to [0, last_win] as one instruction) */
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(w) : "v"(qi), "s"(last_win));
        return (uint32_t)w;
    };
    constexpr uint32_t U = BIN_BLOCK;
    constexpr uint32_t KMAX = (1u << PYA_BIN_KEY_BITS) - 1u;
    w_first[lane] = 0xffffu;                                  /* (no peak) */
    /* only the high word of an intensity is needed here (the keys; sign, infinity and NaN show in it too) */
    const uint32_t *inten_hi = (const uint32_t *)inten + 1;
    uint32_t maxhw = 0;
    int bad = 0;
    /* (a spectrum of more than one block: the later blocks' intensities once more, for the key base) */
    for (uint32_t i = 64 * U + (uint32_t)lane; i < P; i += 64) {
        const uint32_t hw = inten_hi[2 * i];
        bad |= hw >= 0x7ff00000u ? 1 : 0;
        maxhw = hw > maxhw ? hw : maxhw;
    }
    /* m/z order: every peak against the next lane's below (one vector compare per 64 peaks), and the pairs that
     * straddle two chunks here: lane j looks at peaks 64 j + 63 and 64 j + 64 */
    bool straddle = false;
    for (uint32_t j = (uint32_t)lane; 64 * j + 64 < P; j += 64) straddle = straddle || mz[64 * j + 63] > mz[64 * j + 64];
    uint64_t uns = __ballot(straddle);                       /* lanes that saw a larger m/z before a smaller one (scalar) */
    uint32_t carry_w = PYA_BIN_FAST_WINDOWS, keybase = 0;    /* ("window 64" precedes the first peak: its slot takes the -1) */
    /* The sweep that bins the peaks, checks the order and builds the keys.  The spectrum comes in blocks of
     * 64 * BIN_BLOCK peaks whose loads are ALL issued before the first of them is used: a wavefront's time here is
     * HBM round trips, and a load-use-load-use loop makes one per 64 peaks instead of one per block. */
    for (uint32_t base = 0; base < P; base += 64 * U) {
        double v[U];
        uint32_t hw[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = base + u * 64 + (uint32_t)lane;       /* (past the end: the last peak again) */
            const uint32_t ic = i < P ? i : P - 1;
            if (PRE && base == 0) {
                v[u] = pre->v[u];
                hw[u] = pre->hw[u];
            } else {
                v[u] = mz[ic];
                hw[u] = inten_hi[2 * ic];
            }
        }
        if (base == 0) {
            uint32_t m = maxhw;
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {                      /* (clamped lanes repeat the last peak) */
                bad |= hw[u] >= 0x7ff00000u ? 1 : 0;
                m = hw[u] > m ? hw[u] : m;
            }
            m = wave_max_u32(m);
            keybase = m > KMAX ? m - KMAX : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            /* (one chunk at a time: interleaving them costs the registers and the occupancy) */
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t cbase = base + u * 64;
            if (cbase < P) {
                const uint32_t i = cbase + (uint32_t)lane;
                const bool in = i < P;
                const double x = v[u];
                /* (lane 63 sees zeros and is left out: its pair is one of the straddling ones above; the lanes past the
                 * end repeat the last peak) */
                uns |= __ballot(x > lane_next_f64_or_zero(x)) & 0x7fffffffffffffffull;
                const uint32_t w = window_of(x);
                const uint32_t pw = lane_prev_u32(w, carry_w);
                carry_w = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);    /* (ends as the window of the last peak) */
                const uint32_t k = (hw[u] > keybase ? hw[u] : keybase) - keybase;
                if (in) {
                    ckey[i] = ((63u - w) << PYA_BIN_KEY_BITS) | k;
                    s_mzf[i] = (float)x;
                    if (pw != w) {                                /* a window starts here, the one before has ended */
                        w_last[pw] = (uint16_t)(i - 1u);
                        w_first[w] = (uint16_t)i;
                    }
                }
            }
        }
    }
    if (uns) return PYA_BIN_REDO;                            /* peaks out of m/z order */
    if (!ok) {
        *status = nb_f > 65535.f ? PYA_ST_TOO_MANY_BINS : PYA_ST_NO_BINS;
        return -1;
    }
    if (__any(bad) || n_bins > PYA_BIN_FAST_WINDOWS || (b.debug & 128)) return PYA_BIN_REDO;
    /* (the mates of a chunk's peaks are read for as many steps as the longest window reaching into THAT chunk has peaks:
     * a fifth fewer than with the spectrum's longest window for all) */
    uint32_t *cmax = (uint32_t *)(lds + (((size_t)cap * 13 + 63) & ~(size_t)63) + 260);      /* [15], behind the window table */
    const bool per_chunk = P <= 960u;
    if (lane == 0) w_last[carry_w] = (uint16_t)(P - 1u);
    if (lane < 15) cmax[lane] = 0u;
    wave_lds_sync();
    const uint32_t wf = w_first[lane], wl = w_last[lane];
    const uint32_t mylen = wf != 0xffffu ? wl - wf + 1u : 0u;
    const uint32_t maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(mylen));   /* (a scalar trip count) */
    const uint32_t trips = (maxlen + 7u) & ~7u;               /* mates are read eight at a time */
    if (per_chunk && mylen)
        for (uint32_t c = wf >> 6; c <= (wl >> 6); c++) atomicMax(&cmax[c], mylen);
    for (uint32_t q = (uint32_t)lane; q < trips; q += 64) ckey[P + q] = 0u;  /* (index < 2 cap: trips <= cap, a multiple of 32) */
    wave_lds_sync();
    STAMP_T(b, 2, -1);

    /* ranks (Spectra.cpp:24-41) and, in the same sweep, the retained peaks in ascending m/z */
    PeakEntry *dst = DIRECT ? b.ret + b.ret_off[psm] : nullptr;
    uint32_t total = 0;
    int deficit = 0;
    for (uint32_t base = 0; base < P; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in = i < P;
        const uint32_t ic = in ? i : P - 1u;                 /* (past the end: the last peak again, left out below) */
        const uint32_t me = ckey[ic];
        const uint32_t lo = (uint32_t)w_first[63u - (me >> PYA_BIN_KEY_BITS)];
        const float mzf = s_mzf[ic];
        const uint32_t *src = ckey + lo;
        uint32_t c0 = 0, c1 = 0;
        const uint32_t tc = per_chunk ? (((uint32_t)__builtin_amdgcn_readfirstlane((int)cmax[base >> 6]) + 3u) & ~3u) : trips;
        if (!(b.debug & 32))
#pragma unroll 2
        for (uint32_t t = 0; t < tc; t += 4) {                /* (trips is a multiple of 8, a chunk's own count of 4) */
#pragma unroll
            for (uint32_t q = 0; q < 4; q += 2) {
                const uint32_t o0 = src[t + q], o1 = src[t + q + 1];
                c0 += (me - o0) >> 31;                       /* keys are below 2^31: the sign says "more intense" */
                c1 += (me - o1) >> 31;
            }
        }
        const uint32_t cnt = c0 + c1;
        deficit += in ? (int)cnt - (int)(i - lo) : 0;        /* 0 over a window whose keys all differ */
        const bool keep = in && cnt < (uint32_t)ntop;
        const uint64_t m = __ballot(keep);
        if (keep && !(b.debug & 64)) {
            const uint32_t pos = total + lanes_below(m);
            if (DIRECT) {
                PeakEntry e;
                e.mz = mzf;
                e.rank = cnt;
                dst[pos] = e;
            } else {
                s_mzf[pos] = mzf;                            /* pos <= i: in place */
                o_rank[pos] = (uint8_t)cnt;
            }
        }
        total += (uint32_t)__popcll(m);
    }
    if (wave_sum_i32(deficit) != 0 && !(b.debug & 32)) {
        /* Equal keys somewhere: the sweep once more, with the peaks that have such a mate and a count inside the top
         * n_top ordered by their whole intensities (read again from memory: a few lanes, rarely).  Intensities that
         * differ order the peaks strictly, whatever std::nth_element does; equal ones do not -- handed over. */
        bool tie = false;
        total = 0;
        for (uint32_t base = 0; base < P; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool in = i < P;
            const uint32_t me = in ? ckey[i] : 0u;
            const uint32_t w = 63u - (me >> PYA_BIN_KEY_BITS);
            const uint32_t lo = in ? (uint32_t)w_first[w] : 0u;
            const uint32_t *src = ckey + lo;
            uint32_t cnt = 0, eq = 0;
            for (uint32_t t = 0; t < trips; t++) {
                const uint32_t o = src[t];
                cnt += o > me ? 1u : 0u;
                eq += o == me ? 1u : 0u;                     /* (the peak itself included) */
            }
            if (in && eq > 1u && cnt < (uint32_t)ntop) {
                const double mine = inten[i];
                const uint32_t hi = (uint32_t)w_last[w];
                for (uint32_t j = lo; j <= hi; j++) {
                    if (j == i || ckey[j] != me) continue;
                    const double o = inten[j];
                    cnt += o > mine ? 1u : 0u;
                    tie = tie || o == mine;
                }
            }
            const bool keep = in && cnt < (uint32_t)ntop;
            const uint64_t m = __ballot(keep);
            if (keep && !(b.debug & 64)) {
                const uint32_t pos = total + lanes_below(m);
                if (DIRECT) {
                    PeakEntry e;
                    e.mz = (float)mz[i];
                    e.rank = cnt;
                    dst[pos] = e;
                } else {
                    s_mzf[pos] = (float)mz[i];               /* (the first sweep has compacted over this array) */
                    o_rank[pos] = (uint8_t)cnt;
                }
            }
            total += (uint32_t)__popcll(m);
        }
        if (__any(tie)) return PYA_BIN_REDO;
    }
    if (DIRECT) {
        /* what bin_store adds to the entries: the pad behind an odd count, the count, the status */
        if (b.debug & 64) total = 0;
        if (lane == 0) {
            if (total & 1u) {
                PeakEntry e;
                e.mz = __builtin_huge_valf();
                e.rank = (uint32_t)PYA_NO_MATCH;
                dst[total] = e;
            }
            b.ret_n[psm] = total;
            b.status[psm] = PYA_ST_OK;
        }
        STAMP_T(b, 4, -1);
        return (int)total;
    }
    wave_lds_sync();
    STAMP_T(b, 4, -1);
    *out_mz = s_mzf;
    *out_rank = o_rank;
    return (b.debug & 64) ? 0 : (int)total;
}

/* lanes hand data to each other through the arrays of the general body: LDS, or (GLOBAL: spectra of more than 8 192
 * peaks, pya_bin_global_kernel) a scratch area in the workspace */
template <bool GLOBAL>
DEV void bin_sync() {
    if (GLOBAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        wave_lds_sync();
    }
}

/* The general body (see above): any peak order, any number of windows, equal intensities resolved as
 * std::nth_element + std::sort resolve them.  LDS: inten f64[cap] | mzf f32[cap] | window u16[cap] | rank u8[cap]
 * (+ 192 bytes of window starts behind it). */
template <bool GLOBAL>
DEV int bin_exact(const BatchDev &b, uint32_t psm, unsigned char *lds, uint32_t cap, const float **out_mz,
                  const uint8_t **out_rank, int *status) {
    const int lane = lane_id();
    double *s_inten = (double *)lds;
    float *s_mzf = (float *)(s_inten + cap);
    uint16_t *s_bin = (uint16_t *)(s_mzf + cap);
    uint8_t *s_rank = (uint8_t *)(s_bin + cap);

    STAMP_BEGIN();
    STAMP_T(b, 1, -1);
    const int64_t p0 = b.peak_off[psm];
    const int P = (int)(b.peak_off[psm + 1] - p0);
    const double *mz = b.mz + p0;
    const double *inten = b.inten + p0;
    const DevConfig *cfg = b.cfg;
    (void)cfg;

    /* passes 1+2 fused for the normal case: a spectrum sorted by m/z has its extremes at the
     * ends, so the window bounds are computed from mz[0] / mz[P-1] up front and ONE sweep bins
     * the peaks while it checks the order and tracks the true min / max.  If the order check
     * fails (or the ends were not the extremes) the bins are recomputed from the true extremes
     * (Spectra.cpp:46-47 use min_element / max_element). */
    const DevConfig *cfg_ = cfg;
    const float bin_size = cfg_->bin_size;
    const int ntop = cfg_->n_top;
    const double bsd = (double)bin_size, inv_bs = 1. / bsd;
    double mn = mz[0], mx = mz[P - 1];
    float min_mz = 0.f;
    uint32_t n_bins = 0;
    int unsorted = 0;
    *status = PYA_ST_OK;
    /* window bounds from the extremes (Spectra.cpp:46-48); false = no usable number of windows */
    auto bounds = [&](float *nb_out) -> bool {
        min_mz = (float)(__builtin_floor(mn / 100.) * 100.);
        const float max_mz = (float)(__builtin_ceil(mx / 100.) * 100.);
        const float nb_f = __builtin_ceilf((max_mz - min_mz) / bin_size);   /* float arithmetic, :48 */
        *nb_out = nb_f;
        const bool ok = nb_f >= 1.f && nb_f <= 65535.f;
        n_bins = ok ? (uint32_t)nb_f : 1u;
        return ok;
    };
    /* window id of one peak (double arithmetic, Spectra.cpp:55-58): floor((v - min) / bin_size) as the
     * reference's double division gives it, without the division: every multiple k * bin_size
     * (k < 2^16, bin_size a float) is exact in double, so the rounded quotient reaches k exactly when
     * the true one does and the floor equals the mathematical one -- which a reciprocal estimate plus
     * an exact remainder (fma) pins down. */
    auto window_of = [&](double v) -> uint16_t {
        const double x = v - (double)min_mz;
        const double q = __builtin_floor(x * inv_bs);
        const double r = __builtin_fma(-q, bsd, x);
        int qi = (int)q;                                    /* (the +-1 and the clamp as integers: five instructions for ten) */
        qi += r >= bsd ? 1 : 0;
        qi -= r < 0. ? 1 : 0;
        const int last = (int)n_bins - 1;
        return (uint16_t)(qi > last ? last : qi);
    };
    bool ok = true;
    {
        /* The sweep that bins the peaks, checks the order and tracks the true extremes.  The spectrum
         * comes in blocks of 64 * BIN_BLOCK peaks whose loads are ALL issued before the first of them is used: a
         * wavefront's time here is HBM round trips, and a load-use-load-use loop makes one (two with
         * the look at the next peak) per 64 peaks instead of one per block.  The next peak comes from
         * the neighbouring lane, the one after a chunk's last from the next chunk or one extra load. */
        constexpr int U = BIN_BLOCK;
        double tmn = __builtin_huge_val(), tmx = -__builtin_huge_val();
        int uns = 0;
        for (int base = 0; base < P; base += 64 * U) {
            double v[U], it[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + u * 64 + lane;          /* (past the end: the last peak again) */
                const int ic = i < P ? i : P - 1;
                v[u] = mz[ic];
                it[u] = inten[ic];
            }
            const int ie = base + 64 * U;
            const double edge = mz[ie < P ? ie : P - 1];
            if (base == 0) {
                float nb_f;
                ok = bounds(&nb_f);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                /* (one chunk at a time: interleaving the eight costs 140 registers and the occupancy) */
                __builtin_amdgcn_sched_barrier(0);
                if (base + u * 64 < P) {
                    const int i = base + u * 64 + lane;
                    const bool next_chunk = base + (u + 1) * 64 < P;
                    double nx = __shfl_down(v[u], 1, 64);
                    const double after = !next_chunk ? v[u] : (u + 1 < U ? wave_bcast(v[u + 1 < U ? u + 1 : u], 0) : edge);
                    if (lane == 63) nx = after;
                    if (i < P) {
                        const double x = v[u];
                        tmn = x < tmn ? x : tmn;
                        tmx = x > tmx ? x : tmx;
                        uns |= (x > nx) ? 1 : 0;
                        s_inten[i] = it[u];
                        s_mzf[i] = (float)x;
                        s_bin[i] = window_of(x);
                    }
                }
            }
        }
        unsorted = __any(uns);
        bool redo = !ok;
        if (unsorted) {
            tmn = wave_min_f64(tmn);
            tmx = wave_max_f64(tmx);
            redo = redo || tmn != mn || tmx != mx;
            mn = tmn;
            mx = tmx;
        }
        if (redo) {
            /* the ends were not the extremes (or gave no windows): bin again from the true ones */
            float nb_f;
            if (!bounds(&nb_f)) {
                *status = nb_f > 65535.f ? PYA_ST_TOO_MANY_BINS : PYA_ST_NO_BINS;
                return -1;
            }
            for (int i = lane; i < P; i += 64) s_bin[i] = window_of(mz[i]);
        }
    }
    bin_sync<GLOBAL>();
    STAMP_T(b, 2, -1);

    /* pass 3: intensity rank inside the window = number of window mates that are more intense
     * (ties: the earlier peak ranks first; the reference leaves ties unspecified).
     * Sorted spectra (the normal case): a window is a run of consecutive peaks, so each lane gets
     * the bounds [lo, hi] of its run from a ballot of the run starts and all lanes sweep their
     * runs in lock step -- a wave-uniform loop without per-lane exit tests; positions past the end
     * of a shorter run are clamped to the lane's own peak, which never counts. */
    if (!(b.debug & 32)) {
        if (!unsorted) {
            /* Which equally intense peaks of a window are retained, and in which rank order, is
             * whatever std::nth_element + std::sort leave: one lane per window emulates them
             * serially on the window's (intensity, index) pairs in place -- up to 64 windows at a
             * time, so the serial chains of a spectrum's windows run side by side. */
            uint16_t *run_lo = (uint16_t *)(lds + (size_t)cap * 15);   /* [66] window starts of a round */
            int scan = 0;
            while (scan < P) {
                int cnt = 0;
                const int first = scan;
                while (scan < P && cnt <= 64) {                  /* 64 windows and the start of the 65th */
                    const int i = scan + lane;
                    const bool start = i < P && (i == first || s_bin[i] != s_bin[i - 1]);
                    const uint64_t m = __ballot(start);
                    const int here = __popcll(m);
                    if (start) {
                        const int r = cnt + mask_rank(m);
                        if (r <= 64) run_lo[r] = (uint16_t)i;
                    }
                    if (cnt + here > 64) {
                        /* the 65th start (rank 64) closes the round: the next one resumes there */
                        uint64_t mm = m;
                        for (int q = 0; q < 64 - cnt; q++) mm &= mm - 1;
                        scan += __builtin_ctzll(mm);
                        cnt = 64;
                        break;
                    }
                    cnt += here;
                    scan += 64;
                }
                if (scan >= P) {
                    scan = P;
                    if (lane == 0 && cnt <= 64) run_lo[cnt] = (uint16_t)P;   /* the last window ends with the spectrum */
                }
                if (cnt > 64) cnt = 64;
                bin_sync<GLOBAL>();
                if (lane < cnt) {
                    const int lo = run_lo[lane], len = (int)run_lo[lane + 1] - lo;
                    RunDirect r;
                    r.key = s_inten + lo;
                    r.idx = s_bin + lo;
                    for (int e = 0; e < len; e++) r.idx[e] = (uint16_t)e;
                    run_exact_ranks(r, len, s_rank + lo, ntop);
                }
                bin_sync<GLOBAL>();
            }
        } else {
            /* peaks out of m/z order: a window's peaks are scattered, every pair is compared */
            int tie = 0;
            for (int base = 0; base < P; base += 64) {
                const int i = base + lane;
                if (i < P) {
                    const uint16_t w = s_bin[i];
                    const double me = s_inten[i];
                    int cnt = 0;
                    for (int j = 0; j < P; j++) {
                        if (s_bin[j] != w || j == i) continue;
                        const double o = s_inten[j];
                        cnt += (o > me || (o == me && j < i)) ? 1 : 0;
                        tie |= (o == me) ? 1 : 0;
                    }
                    s_rank[i] = (uint8_t)(cnt < ntop ? cnt : 255);
                }
            }
            if (__any(tie)) {
                /* Equal intensities in a window: std::nth_element + std::sort work on the window's
                 * peaks in input order.  A stable counting sort by window id lists them (peak
                 * indices, in the PSM's still unused slice of the retained-m/z output as scratch),
                 * then every window is emulated by the lane of its first list entry. */
                uint16_t *list = (uint16_t *)(b.ret + b.ret_off[psm]);
                for (int base = 0; base < P; base += 64) {
                    const int i = base + lane;
                    if (i < P) {
                        const uint16_t w = s_bin[i];
                        int pos = 0;
                        for (int j = 0; j < P; j++) {
                            const uint16_t wj = s_bin[j];
                            pos += (wj < w || (wj == w && j < i)) ? 1 : 0;
                        }
                        list[pos] = (uint16_t)i;
                    }
                }
                __threadfence();
                bin_sync<GLOBAL>();
                for (int base = 0; base < P; base += 64) {
                    const int q = base + lane;
                    const bool in = q < P;
                    const uint32_t w = in ? (uint32_t)s_bin[list[q]] : 0x10000u;
                    const uint32_t pw = (in && q > 0) ? (uint32_t)s_bin[list[q - 1]] : 0x10001u;
                    const bool start = in && pw != w;
                    const uint64_t starts = __ballot(start);
                    const uint64_t gt = starts & ~(lanemask_lt() | (1ull << lane));
                    int run_end = P - 1;
                    for (int nb = base + 64; nb < P; nb += 64) {
                        const int j = nb + lane;
                        const uint64_t m2 = __ballot(j < P && s_bin[list[j]] != s_bin[list[j - 1]]);
                        if (m2) {
                            run_end = nb + __builtin_ctzll(m2) - 1;
                            break;
                        }
                    }
                    const int hi = gt ? base + __builtin_ctzll(gt) - 1 : run_end;
                    if (start) {
                        RunIndirect r;
                        r.idx = list + q;
                        r.key = s_inten;
                        run_exact_ranks(r, hi - q + 1, s_rank, ntop);
                    }
                }
                __threadfence();
            }
        }
    }
    bin_sync<GLOBAL>();
    STAMP_T(b, 3, -1);

    /* pass 4: retained peaks in ascending float m/z, written over the (no longer needed)
     * intensity array */
    float *o_mz = (float *)s_inten;
    uint8_t *o_rank = (uint8_t *)(o_mz + cap);
    int total = 0;
    if (b.debug & 64) {
    } else if (!unsorted) {
        for (int base = 0; base < P; base += 64) {
            int i = base + lane;
            bool keep = i < P && s_rank[i] < ntop;
            uint64_t m = __ballot(keep);
            if (keep) {
                int pos = total + mask_rank(m);
                o_mz[pos] = s_mzf[i];
                o_rank[pos] = s_rank[i];
            }
            total += __popcll(m);
        }
    } else {
        /* general order: position = number of retained peaks with a smaller (m/z, index) */
        for (int base = 0; base < P; base += 64) {
            int i = base + lane;
            bool keep = i < P && s_rank[i] < ntop;
            if (keep) {
                float me = s_mzf[i];
                int pos = 0;
                for (int j = 0; j < P; j++) {
                    if (s_rank[j] >= ntop) continue;
                    float o = s_mzf[j];
                    pos += (o < me || (o == me && j < i)) ? 1 : 0;
                }
                o_mz[pos] = me;
                o_rank[pos] = s_rank[i];
            }
            total += __popcll(__ballot(keep));
        }
    }
    bin_sync<GLOBAL>();
    STAMP_T(b, 4, -1);
    *out_mz = o_mz;
    *out_rank = o_rank;
    return total;
}

template <bool EXACT>
DEV int bin_core(const BatchDev &b, uint32_t psm, unsigned char *lds, uint32_t cap, const float **out_mz,
                 const uint8_t **out_rank, int *status) {
    if (EXACT) return bin_exact<false>(b, psm, lds, cap, out_mz, out_rank, status);
    return bin_fast<false>(b, psm, lds, cap, out_mz, out_rank, status);
}

/* retained table of one spectrum (or its error status) to global memory */
DEV void bin_store(const BatchDev &b, uint32_t psm, int R, int status, const float *r_mz, const uint8_t *r_rank) {
    const int lane = lane_id();
    if (R < 0) {
        if (lane == 0) {
            b.status[psm] = status;
            b.ret_n[psm] = 0;
        }
        return;
    }
    /* two entries per lane and 16-byte store (the table starts at an even entry: common.h) */
    uint4 *dst = (uint4 *)(b.ret + b.ret_off[psm]);
    const int pairs = (R + 1) >> 1;
    for (int q = lane; q < pairs; q += 64) {
        const int i0 = 2 * q, i1 = 2 * q + 1;
        uint4 v;
        v.x = __float_as_uint(r_mz[i0]);
        v.y = (uint32_t)r_rank[i0];
        v.z = i1 < R ? __float_as_uint(r_mz[i1]) : __float_as_uint(__builtin_huge_valf());
        v.w = i1 < R ? (uint32_t)r_rank[i1] : (uint32_t)PYA_NO_MATCH;
        dst[q] = v;
    }
    if (lane == 0) {
        b.ret_n[psm] = (uint32_t)R;
        b.status[psm] = PYA_ST_OK;
    }
}

#endif
