/* bin_spectra.hip -- kernel 1: raw peaks -> retained-peak table (one spectrum per wavefront).
 *
 * Replaces BinnedSpectra::consumeSpectra + sortTopSpectra (cpp/Spectra.cpp:43-68, :24-41) and
 * the (bin, rank) peak feed of Ascore.pyx:142-150.  Output per spectrum: the retained peaks
 * (the n_top = 10 most intense of every bin_size-wide window) as float32 m/z in ascending
 * order plus their intensity rank inside their window -- exactly the information the match
 * cache of the reference holds (ModifiedPeptide.cpp:126-142).
 *
 * HBM traffic: reads 16 B per raw peak once (coalesced, 8 B per lane), writes 5 B per retained
 * peak.  Everything else lives in LDS: intensity (f64) and window id (u16) per peak.
 */
#include "device_common.hip.h"

struct BinLds {
    double *inten;      /* [cap] */
    float *mzf;         /* [cap] */
    uint16_t *bin;      /* [cap] */
    uint8_t *rank;      /* [cap] */
};

__global__ __launch_bounds__(64) void pya_bin_spectra_kernel(BatchDev b, uint32_t n_psm, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t psm = blockIdx.x;
    if (psm >= n_psm) return;
    const int lane = lane_id();
    BinLds s;
    s.inten = (double *)lds_raw;
    s.mzf = (float *)(s.inten + cap);
    s.bin = (uint16_t *)(s.mzf + cap);
    s.rank = (uint8_t *)(s.bin + cap);

    const int64_t p0 = b.peak_off[psm];
    const int P = (int)(b.peak_off[psm + 1] - p0);
    const double *mz = b.mz + p0;
    const double *inten = b.inten + p0;
    const DevConfig *cfg = b.cfg;

    /* pass 1: min / max / sortedness (Spectra.cpp:46-47 use min_element / max_element) */
    double mn = __builtin_huge_val(), mx = -__builtin_huge_val();
    int unsorted = 0;
    for (int i = lane; i < P; i += 64) {
        double v = mz[i];
        double nx = (i + 1 < P) ? mz[i + 1] : v;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
        unsorted |= (v > nx) ? 1 : 0;
    }
    mn = wave_min_f64(mn);
    mx = wave_max_f64(mx);
    unsorted = __any(unsorted);

    const float min_mz = (float)(__builtin_floor(mn / 100.) * 100.);
    const float max_mz = (float)(__builtin_ceil(mx / 100.) * 100.);
    const float bin_size = cfg->bin_size;
    const float nb_f = __builtin_ceilf((max_mz - min_mz) / bin_size);   /* float arithmetic, :48 */
    int status = PYA_ST_OK;
    if (!(nb_f >= 1.f)) status = PYA_ST_NO_BINS;
    if (nb_f > 65535.f) status = PYA_ST_TOO_MANY_BINS;
    if (status != PYA_ST_OK) {
        if (lane == 0) {
            b.status[psm] = status;
            b.ret_n[psm] = 0;
        }
        return;
    }
    const uint32_t n_bins = (uint32_t)nb_f;

    /* pass 2: window id per peak (double arithmetic, Spectra.cpp:55-58) */
    for (int i = lane; i < P; i += 64) {
        double v = mz[i];
        double q = __builtin_floor((v - (double)min_mz) / (double)bin_size);
        uint32_t w = q >= (double)(n_bins - 1) ? n_bins - 1 : (uint32_t)q;
        s.bin[i] = (uint16_t)w;
        s.inten[i] = inten[i];
        s.mzf[i] = (float)v;
    }
    wave_lds_sync();

    /* pass 3: intensity rank inside the window = number of window mates that are more intense
     * (ties: the earlier peak ranks first; the reference leaves ties unspecified). */
    for (int base = 0; base < P; base += 64) {
        int i = base + lane;
        int cnt = PYA_NTOP;
        if (i < P) {
            const uint16_t w = s.bin[i];
            const double me = s.inten[i];
            cnt = 0;
            if (!unsorted) {
                for (int j = i - 1; j >= 0 && cnt < PYA_NTOP && s.bin[j] == w; j--)
                    cnt += (s.inten[j] >= me) ? 1 : 0;
                for (int j = i + 1; j < P && cnt < PYA_NTOP && s.bin[j] == w; j++)
                    cnt += (s.inten[j] > me) ? 1 : 0;
            } else {
                for (int j = 0; j < P && cnt < PYA_NTOP; j++) {
                    if (s.bin[j] != w || j == i) continue;
                    double o = s.inten[j];
                    cnt += (o > me || (o == me && j < i)) ? 1 : 0;
                }
            }
        }
        if (i < P) s.rank[i] = (uint8_t)(cnt < PYA_NTOP ? cnt : PYA_NO_MATCH);
    }
    wave_lds_sync();

    /* pass 4: emit retained peaks in ascending float m/z */
    float *out_mz = b.ret_mz + p0;
    uint8_t *out_rank = b.ret_rank + p0;
    int total = 0;
    if (!unsorted) {
        for (int base = 0; base < P; base += 64) {
            int i = base + lane;
            bool keep = i < P && s.rank[i] < PYA_NTOP;
            uint64_t m = __ballot(keep);
            if (keep) {
                int pos = total + __popcll(m & lanemask_lt());
                out_mz[pos] = s.mzf[i];
                out_rank[pos] = s.rank[i];
            }
            total += __popcll(m);
        }
    } else {
        /* general order: position = number of retained peaks with a smaller (m/z, index) */
        for (int base = 0; base < P; base += 64) {
            int i = base + lane;
            bool keep = i < P && s.rank[i] < PYA_NTOP;
            if (keep) {
                float me = s.mzf[i];
                int pos = 0;
                for (int j = 0; j < P; j++) {
                    if (s.rank[j] >= PYA_NTOP) continue;
                    float o = s.mzf[j];
                    pos += (o < me || (o == me && j < i)) ? 1 : 0;
                }
                out_mz[pos] = me;
                out_rank[pos] = s.rank[i];
            }
            total += __popcll(__ballot(keep));
        }
    }
    if (lane == 0) {
        b.ret_n[psm] = (uint32_t)total;
        b.status[psm] = PYA_ST_OK;
    }
}

extern "C" size_t pya_bin_lds_bytes(uint32_t cap) {
    return (size_t)cap * (8 + 4 + 2 + 1) + 64;
}

extern "C" int pya_launch_bin(const BatchDev *b, uint32_t n_psm, uint32_t cap, hipStream_t stream) {
    if (n_psm == 0) return 0;
    /* cap rounded so that every LDS sub-array stays aligned */
    hipLaunchKernelGGL(pya_bin_spectra_kernel, dim3(n_psm), dim3(64), pya_bin_lds_bytes(cap), stream,
                       *b, n_psm, cap);
    return (int)hipGetLastError();
}
