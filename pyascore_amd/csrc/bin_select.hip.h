/* bin_select.hip.h -- binning of DENSE spectra (one wavefront per spectrum): selection first, ranks afterwards.
 *
 * BinnedSpectra::sortTopSpectra (cpp/Spectra.cpp:24-41) is std::nth_element + a sort of the n_top survivors: O(window) per
 * window.  bin_fast (bin_core.hip.h) ranks every peak against every mate of its window -- O(window^2), and 13 bytes of LDS
 * per raw peak: right for the ~17 peaks per window of a sparse spectrum (cfg2: 323 peaks), 20 x the time per peak at 4 000
 * peaks (214 per window; r06 bench legs dense1500 / dense4000).  Here:
 *   pass 0  order / sign / finiteness checks and the largest intensity (the key base), as bin_fast makes them;
 *   pass 1  a histogram per window over the top 6 bits of the 25-bit intensity key (half an octave per bucket): LDS atomics,
 *           64 windows x 64 buckets of 16 bits;
 *   thresholds  per window the highest bucket b with  #{peaks of the window in buckets >= b} >= n_top  (0 when the window has
 *           fewer peaks): every peak of the window's top n_top lies in a bucket >= b, and so does every peak MORE INTENSE
 *           than any such peak -- the survivors are closed upwards;
 *   pass 1b only when the survivors so chosen would not fit their slots (intensities packed into few buckets: a flat noise
 *           floor, a narrow dynamic range): a second histogram, over the NEXT 6 bits, of the peaks in their window's threshold
 *           bucket, and per window the highest sub-bucket that still leaves n_top peaks at or above (bucket, sub-bucket):
 *           the survivors are the peaks whose top 12 key bits are >= that pair -- still closed upwards, still at least n_top
 *           per window, now about one percent of intensity wide at the threshold instead of a factor of 1.4;
 *   pass 2  the survivors (bucket >= threshold of their window: n_top plus what shares the threshold bucket, ~ 15 per window)
 *           compacted in m/z order into LDS: composite key, float m/z, index of the raw peak;
 *   ranks   bin_fast's sweep over the survivors only.  A survivor's rank among the survivors of its window IS its rank in the
 *           window (everything more intense survived), equal keys share a bucket (all of them survive or none), so the
 *           deficit test, the second sweep by whole intensities and the hand-over of truly equal intensities carry over
 *           unchanged.
 * The raw spectrum is read three times (four with pass 1b; all but the first from L2 / the Infinity Cache) and never staged: LDS is max(8 KB, 14 B per survivor slot) per
 * wavefront whatever the peak count.  More survivors than slots (flat intensities: everything in one bucket), peaks out of
 * order, more than 64 windows, bad intensities, equal intensities inside a top n_top: PYA_BIN_REDO, as bin_fast answers.
 * Results are stored straight to the workspace table (the batch kernel's DIRECT way). */
#ifndef PYA_BIN_SELECT_H
#define PYA_BIN_SELECT_H
#include "bin_core.hip.h"

#define PYA_BIN_SEL_HIST_BYTES (PYA_BIN_FAST_WINDOWS * 32 * 4)      /* [window][32 words of two 16-bit buckets] */
#define PYA_BIN_SEL_TAIL 640                                           /* thresholds (two levels), window table, chunk maxima */
__host__ __device__ static inline size_t pya_bin_sel_area(uint32_t scap) {
    const size_t s = (((size_t)scap * 14 + 63) & ~(size_t)63);
    return s > PYA_BIN_SEL_HIST_BYTES ? s : PYA_BIN_SEL_HIST_BYTES;
}
#define PYA_BIN_SEL_BYTES(scap) (pya_bin_sel_area(scap) + PYA_BIN_SEL_TAIL)

DEV int bin_select(const BatchDev &b, uint32_t psm, unsigned char *lds, uint32_t scap, int *status) {
    const int lane = lane_id();
    uint32_t *hist = (uint32_t *)lds;
    uint32_t *ckey = (uint32_t *)lds;                        /* [2 scap] survivors' keys, then zeros for the longest window */
    float *s_mzf = (float *)(ckey + 2 * (size_t)scap);       /* [scap] */
    uint16_t *s_idx = (uint16_t *)(s_mzf + scap);            /* [scap] raw peak of a survivor */
    unsigned char *tail = lds + pya_bin_sel_area(scap);
    uint8_t *thr = tail;                                     /* [64] threshold bucket per window */
    uint16_t *w_last = (uint16_t *)(tail + 64);              /* [64] + the slot of "window 64" */
    uint16_t *w_first = w_last + PYA_BIN_FAST_WINDOWS + 1;
    uint32_t *cmax = (uint32_t *)(tail + 64 + 264);          /* [15] */
    uint8_t *thr2 = tail + 448;                              /* [64] threshold sub-bucket per window (0: no second level) */
    uint8_t *need = tail + 512;                              /* [64] peaks still wanted from the window's threshold bucket */

    STAMP_BEGIN();
    STAMP_T(b, 1, -1);
    const int64_t p0 = b.peak_off[psm];
    const uint32_t P = (uint32_t)(b.peak_off[psm + 1] - p0);
    const double *mz = b.mz + p0;
    const double *inten = b.inten + p0;
    const uint32_t *inten_hi = (const uint32_t *)inten + 1;
    const float bin_size = b.cfg->bin_size;
    const int ntop = b.cfg->n_top;
    const double bsd = (double)bin_size, inv_bs = __builtin_amdgcn_rcp(bsd);
    const double mn = mz[0], mx = mz[P - 1];
    *status = PYA_ST_OK;
    /* window bounds and ids: bin_fast's arithmetic (Spectra.cpp:46-48, :55-58 without the divisions) */
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
    const float nb_f = __builtin_ceilf((max_mz - min_mz) / bin_size);
    const bool ok = nb_f >= 1.f && nb_f <= 65535.f;
    const uint32_t n_bins = ok ? (uint32_t)nb_f : 1u;
    const int last_win = (int)(n_bins < PYA_BIN_FAST_WINDOWS ? n_bins : PYA_BIN_FAST_WINDOWS) - 1;
    auto window_of = [&](double v) -> uint32_t {
        const double x = v - (double)min_mz;
        const double q = __builtin_floor(x * inv_bs);
        const double r = __builtin_fma(-q, bsd, x);
        int qi = (int)q;
        qi += r >= bsd ? 1 : 0;
        qi -= r < 0. ? 1 : 0;
        int w;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(w) : "v"(qi), "s"(last_win));
        return (uint32_t)w;
    };
    constexpr uint32_t U = BIN_BLOCK;
    constexpr uint32_t KMAX = (1u << PYA_BIN_KEY_BITS) - 1u;
    constexpr uint32_t DSHIFT = PYA_BIN_KEY_BITS - 6;

    /* ---- pass 0: checks and the key base.  Blocks of 64 U peaks, all loads of a block in flight together ---- */
    for (uint32_t i = (uint32_t)lane; i < PYA_BIN_SEL_HIST_BYTES / 4; i += 64) hist[i] = 0u;
    uint32_t maxhw = 0;
    int bad = 0;
    bool straddle = false;
    for (uint32_t j = (uint32_t)lane; 64 * j + 64 < P; j += 64) straddle = straddle || mz[64 * j + 63] > mz[64 * j + 64];
    uint64_t uns = __ballot(straddle);
    for (uint32_t base = 0; base < P; base += 64 * U) {
        double v[U];
        uint32_t hw[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = base + u * 64 + (uint32_t)lane;
            const uint32_t ic = i < P ? i : P - 1;
            v[u] = mz[ic];
            hw[u] = inten_hi[2 * ic];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            bad |= hw[u] >= 0x7ff00000u ? 1 : 0;
            maxhw = hw[u] > maxhw ? hw[u] : maxhw;
            /* (lane 63's pair is one of the straddling ones above; lanes past the end repeat the last peak) */
            uns |= __ballot(v[u] > lane_next_f64_or_zero(v[u])) & 0x7fffffffffffffffull;
        }
    }
    if (uns) return PYA_BIN_REDO;
    if (!ok) {
        *status = nb_f > 65535.f ? PYA_ST_TOO_MANY_BINS : PYA_ST_NO_BINS;
        return -1;
    }
    if (__any(bad) || n_bins > PYA_BIN_FAST_WINDOWS || (b.debug & 128)) return PYA_BIN_REDO;
    maxhw = wave_max_u32(maxhw);
    const uint32_t keybase = maxhw > KMAX ? maxhw - KMAX : 0u;
    wave_lds_sync();
    STAMP_T(b, 2, -1);

    /* ---- pass 1: histograms ---- */
    for (uint32_t base = 0; base < P; base += 64 * U) {
        double v[U];
        uint32_t hw[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = base + u * 64 + (uint32_t)lane;
            const uint32_t ic = i < P ? i : P - 1;
            v[u] = mz[ic];
            hw[u] = inten_hi[2 * ic];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t i = base + u * 64 + (uint32_t)lane;
            if (base + u * 64 < P) {
                const uint32_t w = window_of(v[u]);
                const uint32_t k = (hw[u] > keybase ? hw[u] : keybase) - keybase;
                const uint32_t d = k >> DSHIFT;
                if (i < P) atomicAdd(&hist[w * 32u + (d >> 1)], 1u << ((d & 1u) * 16u));
            }
        }
    }
    wave_lds_sync();
    /* ---- thresholds: a window at a time, a bucket per lane ---- */
    uint32_t s_est = 0;                                                  /* survivors the first level would leave */
    for (int w = 0; w <= last_win; w++) {
        const uint32_t c = (hist[(uint32_t)w * 32u + ((uint32_t)lane >> 1)] >> (((uint32_t)lane & 1u) * 16u)) & 0xffffu;
        const uint32_t incl = wave_incl_scan_u32<false>(c);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t suffix = tot - incl + c;                          /* peaks of the window in buckets >= lane */
        const uint64_t m = __ballot(suffix >= (uint32_t)ntop);
        const int t = m ? 63 - __builtin_clzll(m) : 0;
        const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)suffix, t);          /* ... in buckets >= t */
        const uint32_t in_t = (uint32_t)__builtin_amdgcn_readlane((int)c, t);
        s_est += at;
        if (lane == 0) {
            thr[w] = (uint8_t)t;
            thr2[w] = 0;
            /* peaks above the threshold bucket: at - in_t < n_top (or the window has fewer than n_top peaks: all survive) */
            const uint32_t above = at - in_t;
            need[w] = (uint8_t)(above < (uint32_t)ntop ? (uint32_t)ntop - above : 0u);
        }
    }
    wave_lds_sync();
    if (s_est > scap) {
        /* ---- pass 1b: the threshold buckets, six bits finer ---- */
        constexpr uint32_t DSHIFT2 = DSHIFT - 6;
        for (uint32_t i = (uint32_t)lane; i < PYA_BIN_SEL_HIST_BYTES / 4; i += 64) hist[i] = 0u;
        wave_lds_sync();
        for (uint32_t base = 0; base < P; base += 64 * U) {
            double v[U];
            uint32_t hw[U];
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                const uint32_t i = base + u * 64 + (uint32_t)lane;
                const uint32_t ic = i < P ? i : P - 1;
                v[u] = mz[ic];
                hw[u] = inten_hi[2 * ic];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; u++) {
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t i = base + u * 64 + (uint32_t)lane;
                if (base + u * 64 < P) {
                    const uint32_t w = window_of(v[u]);
                    const uint32_t k = (hw[u] > keybase ? hw[u] : keybase) - keybase;
                    const uint32_t sub = (k >> DSHIFT2) & 63u;
                    if (i < P && (k >> DSHIFT) == (uint32_t)thr[w]) atomicAdd(&hist[w * 32u + (sub >> 1)], 1u << ((sub & 1u) * 16u));
                }
            }
        }
        wave_lds_sync();
        for (int w = 0; w <= last_win; w++) {
            const uint32_t c = (hist[(uint32_t)w * 32u + ((uint32_t)lane >> 1)] >> (((uint32_t)lane & 1u) * 16u)) & 0xffffu;
            const uint32_t incl = wave_incl_scan_u32<false>(c);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t suffix = tot - incl + c;                      /* peaks of the threshold bucket in sub-buckets >= lane */
            const uint32_t nd = (uint32_t)need[w];
            const uint64_t m = __ballot(nd != 0u && suffix >= nd);
            if (lane == 0) thr2[w] = (uint8_t)(m ? 63 - __builtin_clzll(m) : 0);
        }
        wave_lds_sync();
    }
    /* (the histograms are dead: the survivors take their place) */
    STAMP_T(b, 3, -1);

    /* ---- pass 2: the survivors, compacted in m/z order ---- */
    uint32_t S = 0;
    for (uint32_t base = 0; base < P; base += 64 * U) {
        double v[U];
        uint32_t hw[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            const uint32_t i = base + u * 64 + (uint32_t)lane;
            const uint32_t ic = i < P ? i : P - 1;
            v[u] = mz[ic];
            hw[u] = inten_hi[2 * ic];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t i = base + u * 64 + (uint32_t)lane;
            if (base + u * 64 < P) {
                const uint32_t w = window_of(v[u]);
                const uint32_t k = (hw[u] > keybase ? hw[u] : keybase) - keybase;
                /* the top 12 key bits against (threshold bucket, threshold sub-bucket): one compare */
                const bool sv = i < P && (k >> (DSHIFT - 6)) >= (((uint32_t)thr[w] << 6) | (uint32_t)thr2[w]);
                const uint64_t m = __ballot(sv);
                const uint32_t pos = S + lanes_below(m);
                if (sv && pos < scap) {
                    ckey[pos] = ((63u - w) << PYA_BIN_KEY_BITS) | k;
                    s_mzf[pos] = (float)v[u];
                    s_idx[pos] = (uint16_t)i;
                }
                S += (uint32_t)__popcll(m);
            }
        }
    }
    if (S > scap) return PYA_BIN_REDO;                                  /* (flat intensities: more survivors than slots) */
    wave_lds_sync();
    /* the survivors' window table, as bin_fast's first sweep leaves it for the raw peaks */
    w_first[lane] = 0xffffu;
    wave_lds_sync();
    uint32_t carry_w = PYA_BIN_FAST_WINDOWS;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in = i < S;
        const uint32_t w = 63u - (ckey[in ? i : S - 1u] >> PYA_BIN_KEY_BITS);
        const uint32_t pw = lane_prev_u32(w, carry_w);
        carry_w = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
        if (in && pw != w) {
            w_last[pw] = (uint16_t)(i - 1u);
            w_first[w] = (uint16_t)i;
        }
    }
    const bool per_chunk = S <= 960u;
    if (lane == 0) w_last[carry_w] = (uint16_t)(S - 1u);
    if (lane < 15) cmax[lane] = 0u;
    wave_lds_sync();
    const uint32_t wf = w_first[lane], wl = w_last[lane];
    const uint32_t mylen = wf != 0xffffu ? wl - wf + 1u : 0u;
    const uint32_t maxlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(mylen));
    const uint32_t trips = (maxlen + 7u) & ~7u;
    if (per_chunk && mylen)
        for (uint32_t c = wf >> 6; c <= (wl >> 6); c++) atomicMax(&cmax[c], mylen);
    for (uint32_t q = (uint32_t)lane; q < trips; q += 64) ckey[S + q] = 0u;      /* (index < 2 scap) */
    wave_lds_sync();
    STAMP_T(b, 4, -1);

    /* ---- ranks among the survivors (Spectra.cpp:24-41) and the retained peaks, straight to the workspace table ---- */
    PeakEntry *dst = b.ret + b.ret_off[psm];
    uint32_t total = 0;
    int deficit = 0;
    for (uint32_t base = 0; base < S; base += 64) {
        const uint32_t i = base + (uint32_t)lane;
        const bool in = i < S;
        const uint32_t ic = in ? i : S - 1u;
        const uint32_t me = ckey[ic];
        const uint32_t lo = (uint32_t)w_first[63u - (me >> PYA_BIN_KEY_BITS)];
        const float mzf = s_mzf[ic];
        const uint32_t *src = ckey + lo;
        uint32_t c0 = 0, c1 = 0;
        const uint32_t tc = per_chunk ? (((uint32_t)__builtin_amdgcn_readfirstlane((int)cmax[base >> 6]) + 3u) & ~3u) : trips;
#pragma unroll 2
        for (uint32_t t = 0; t < tc; t += 4) {
#pragma unroll
            for (uint32_t q = 0; q < 4; q += 2) {
                const uint32_t o0 = src[t + q], o1 = src[t + q + 1];
                c0 += (me - o0) >> 31;
                c1 += (me - o1) >> 31;
            }
        }
        const uint32_t cnt = c0 + c1;
        deficit += in ? (int)cnt - (int)(i - lo) : 0;
        const bool keep = in && cnt < (uint32_t)ntop;
        const uint64_t m = __ballot(keep);
        if (keep) {
            PeakEntry e;
            e.mz = mzf;
            e.rank = cnt;
            dst[total + lanes_below(m)] = e;
        }
        total += (uint32_t)__popcll(m);
    }
    if (wave_sum_i32(deficit) != 0) {
        /* equal keys among the survivors: the sweep once more, whole intensities deciding (bin_fast's second sweep) */
        bool tie = false;
        total = 0;
        for (uint32_t base = 0; base < S; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            const bool in = i < S;
            const uint32_t me = in ? ckey[i] : 0u;
            const uint32_t w = 63u - (me >> PYA_BIN_KEY_BITS);
            const uint32_t lo = in ? (uint32_t)w_first[w] : 0u;
            const uint32_t *src = ckey + lo;
            uint32_t cnt = 0, eq = 0;
            for (uint32_t t = 0; t < trips; t++) {
                const uint32_t o = src[t];
                cnt += o > me ? 1u : 0u;
                eq += o == me ? 1u : 0u;
            }
            if (in && eq > 1u && cnt < (uint32_t)ntop) {
                const double mine = inten[s_idx[i]];
                const uint32_t hi = (uint32_t)w_last[w];
                for (uint32_t j = lo; j <= hi; j++) {
                    if (j == i || ckey[j] != me) continue;
                    const double o = inten[s_idx[j]];
                    cnt += o > mine ? 1u : 0u;
                    tie = tie || o == mine;
                }
            }
            const bool keep = in && cnt < (uint32_t)ntop;
            const uint64_t m = __ballot(keep);
            if (keep) {
                PeakEntry e;
                e.mz = s_mzf[i];
                e.rank = cnt;
                dst[total + lanes_below(m)] = e;
            }
            total += (uint32_t)__popcll(m);
        }
        if (__any(tie)) return PYA_BIN_REDO;
    }
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
    STAMP_T(b, 5, -1);
    return (int)total;
}

#endif
