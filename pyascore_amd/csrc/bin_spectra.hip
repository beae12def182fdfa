/* bin_spectra.hip -- kernel 1: raw peaks -> retained-peak table (one spectrum per wavefront).
 *
 * Replaces BinnedSpectra::consumeSpectra + sortTopSpectra (cpp/Spectra.cpp:43-68, :24-41) and
 * the (bin, rank) peak feed of Ascore.pyx:142-150.  Output per spectrum: the retained peaks
 * (the n_top = 10 most intense of every bin_size-wide window) as float32 m/z in ascending
 * order plus their intensity rank inside their window -- exactly the information the match
 * cache of the reference holds (ModifiedPeptide.cpp:126-142).
 *
 * HBM traffic: reads 16 B per raw peak once (coalesced, 8 B per lane), writes 8 B per retained
 * peak.  Everything else lives in LDS (bin_core.hip.h).
 *
 * Two kernels: pya_bin_spectra_kernel takes the common case (peaks in m/z order, no two equal
 * intensities inside a window) and appends every other spectrum to a list that
 * pya_bin_exact_kernel works off -- any peak order, ties resolved as std::nth_element + std::sort
 * resolve them.  The rare paths stay out of the hot kernel's register budget that way.
 */
#include "bin_core.hip.h"
#include "bin_select.hip.h"

#ifndef BIN_WAVES
#define BIN_WAVES 4     /* independent spectra per workgroup when LDS allows (no cross-wave sync) */
#endif

__global__ __launch_bounds__(64 * BIN_WAVES) void pya_bin_spectra_kernel(BatchDev b, const uint32_t *psm_ids,
                                                                         uint32_t n_ids, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    /* (the wavefront's number as a scalar: the spectrum's offsets, length and pointers then live in scalar registers and
     * the tests on them are scalar branches) */
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = xcd_slot(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wave;
    /* the plan's runs alternate between two sets of hand-over counts: this run zeroes the next one's (nothing of this
     * run touches them, everything of the run before has finished) -- no memset between the runs */
    if (blockIdx.x == 0 && threadIdx.x < 3 && b.zero_next) b.zero_next[threadIdx.x] = 0u;
    if (slot >= n_ids) return;
    unsigned char *lds_raw = lds_all + (size_t)wave * PYA_BIN_FAST_BYTES(cap);
    const uint32_t psm = psm_ids[slot];
    const float *r_mz;
    const uint8_t *r_rank;
    int status;
    const int R = bin_fast<true>(b, psm, lds_raw, cap, &r_mz, &r_rank, &status);   /* (stores the table itself) */
    if (R == PYA_BIN_REDO) {
        /* peaks out of m/z order or equal intensities in a window: left to pya_bin_exact_kernel */
        if (lane_id() == 0) b.redo_ids[atomicAdd(b.redo_count, 1u)] = psm;
        return;
    }
    if (R < 0) bin_store(b, psm, R, status, r_mz, r_rank);                        /* (no windows: the status only) */
}

/* Dense spectra (the host sends the peak classes above pya_handle.kn.bin_select_min here): selection by per-window
 * histograms, then bin_fast's ranking over the survivors only (bin_select.hip.h) -- O(peaks), LDS independent of the peak
 * count.  What it declines goes to the same list. */
__global__ __launch_bounds__(64 * BIN_WAVES) void pya_bin_select_kernel(BatchDev b, const uint32_t *psm_ids, uint32_t n_ids,
                                                                        uint32_t scap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t slot = xcd_slot(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + wave;
    if (blockIdx.x == 0 && threadIdx.x < 3 && b.zero_next) b.zero_next[threadIdx.x] = 0u;      /* (as pya_bin_spectra_kernel) */
    if (slot >= n_ids) return;
    unsigned char *lds_raw = lds_all + (size_t)wave * PYA_BIN_SEL_BYTES(scap);
    const uint32_t psm = psm_ids[slot];
    int status;
    const int R = bin_select(b, psm, lds_raw, scap, &status);
    if (R == PYA_BIN_REDO) {
        if (lane_id() == 0) b.redo_ids[atomicAdd(b.redo_count, 1u)] = psm;
        return;
    }
    if (R < 0) bin_store(b, psm, R, status, nullptr, nullptr);
}

/* the spectra the kernels above declined, one per wavefront, a fixed grid striding over the list */
__global__ __launch_bounds__(64) void pya_bin_exact_kernel(BatchDev b, uint32_t cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    const uint32_t n = *b.redo_count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t psm = b.redo_ids[k];
        const float *r_mz;
        const uint8_t *r_rank;
        int status;
        const int R = bin_core<true>(b, psm, lds_all, cap, &r_mz, &r_rank, &status);
        bin_store(b, psm, R, status, r_mz, r_rank);
        wave_lds_sync();
    }
}

/* Spectra of more than 8 192 peaks (up to 65 535: 16-bit peak indices): the general body with its arrays in a scratch
 * area of the workspace instead of LDS, one spectrum per wavefront.  Slow (every array access is a trip to memory)
 * and rare; the PSMs behind such spectra are scored by the general kernel, which reads the retained table from the
 * workspace like this kernel leaves it. */
__global__ __launch_bounds__(64) void pya_bin_global_kernel(BatchDev b, const uint32_t *ids, uint32_t n_ids, unsigned char *scratch,
                                                            uint64_t stride, uint32_t cap) {
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = ids[blockIdx.x];
    const float *r_mz;
    const uint8_t *r_rank;
    int status;
    const int R = bin_exact<true>(b, psm, scratch + (size_t)blockIdx.x * stride, cap, &r_mz, &r_rank, &status);
    bin_store(b, psm, R, status, r_mz, r_rank);
}

extern "C" size_t pya_bin_global_scratch_bytes(uint32_t cap) { return PYA_BIN_WAVE_BYTES(cap); }

extern "C" int pya_launch_bin_global(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch,
                                     uint64_t stride, uint32_t cap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    hipLaunchKernelGGL(pya_bin_global_kernel, dim3(n_ids), dim3(64), 0, stream, *b, d_ids, n_ids, d_scratch, stride, cap);
    return (int)hipGetLastError();
}

extern "C" size_t pya_bin_lds_bytes(uint32_t cap) { return PYA_BIN_FAST_BYTES(cap); }

extern "C" int pya_launch_bin(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                              hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t per_wave = PYA_BIN_FAST_BYTES(cap);
    const uint32_t nw = per_wave * BIN_WAVES <= 64 * 1024 ? BIN_WAVES : 1;
    size_t lds = nw * per_wave;
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_bin_spectra_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_bin_spectra_kernel, dim3((n_ids + nw - 1) / nw), dim3(64 * nw), lds, stream, *b, d_ids,
                       n_ids, cap);
    return (int)hipGetLastError();
}

extern "C" size_t pya_bin_select_lds_bytes(uint32_t scap) { return PYA_BIN_SEL_BYTES(scap); }

/* scap: survivor slots per spectrum (a multiple of 32) */
extern "C" int pya_launch_bin_select(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t scap, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t per_wave = PYA_BIN_SEL_BYTES(scap);
    const uint32_t nw = per_wave * BIN_WAVES <= 64 * 1024 ? BIN_WAVES : 1;
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_bin_select_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_bin_select_kernel, dim3((n_ids + nw - 1) / nw), dim3(64 * nw), nw * per_wave, stream, *b, d_ids, n_ids, scap);
    return (int)hipGetLastError();
}

/* after every pya_launch_bin of a batch: the spectra they declined (b->redo_count must have been
 * zeroed before the first of them) */
extern "C" int pya_launch_bin_exact(const BatchDev *b, uint32_t n_total, uint32_t cap, hipStream_t stream) {
    if (n_total == 0) return 0;
    const size_t per_wave = PYA_BIN_WAVE_BYTES(cap);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_bin_exact_kernel);
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = n_total < 16384u ? n_total : 16384u;   /* all spectra may need it (count-like intensities) */
    hipLaunchKernelGGL(pya_bin_exact_kernel, dim3(grid), dim3(64), per_wave, stream, *b, cap);
    return (int)hipGetLastError();
}
