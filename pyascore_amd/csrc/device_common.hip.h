/* device_common.hip.h -- gfx950 device helpers shared by the three kernels.
 *
 * Arithmetic contract (DESIGN.md "Exactness"): everything that decides a match is computed in
 * the reference's own operation order and precision; this translation unit is compiled with
 * -ffp-contract=off and without fast-math so no operation is fused, reassociated or
 * approximated.  Wavefront = 64 lanes throughout; one PSM per wavefront.
 */
#ifndef PYA_DEVICE_COMMON_H
#define PYA_DEVICE_COMMON_H

#include <hip/hip_runtime.h>
#include <atomic>
#include "common.h"

/* Dynamic LDS above 64 KB has to be allowed per kernel.  The allowance is raised ONCE per kernel
 * and device to the CU's 160 KB (it does not influence occupancy; the launch's own size does), so
 * host threads driving different handles never lower each other's limit between the call and the
 * launch. */
static inline hipError_t pya_set_max_lds(const void *fn, std::atomic<uint32_t> &done_mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t bit = 1u << (dev & 31);
    if (done_mask.load(std::memory_order_acquire) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done_mask.fetch_or(bit, std::memory_order_release);
    return e;
}
#define PYA_ENSURE_MAX_LDS(kernel)                            \
    ([]() -> hipError_t {                                     \
        static std::atomic<uint32_t> done_mask_{0};           \
        return pya_set_max_lds((const void *)(kernel), done_mask_); \
    }())

#define DEV __device__ __forceinline__

/* In-kernel phase stamps, diagnostic build only (-DPYA_STAMPS): shares, not lengths. */
#ifdef PYA_STAMPS
#define STAMP_BEGIN() unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime()
#define STAMP(b, k)                                                                  \
    do {                                                                             \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        if ((b).stamps && (threadIdx.x & 63) == 0) atomicAdd(&(b).stamps[k], now_ - stamp_t_); \
        stamp_t_ = __builtin_amdgcn_s_memtime();                                     \
    } while (0)
/* truncation profile: with PYA_DEBUG=k<<16 every wave stops at stamp k (results are garbage; the kernel's
 * duration as a function of k is a cumulative time profile that does not depend on where s_memtime lands) */
#define STAMP_T(b, k, ret)                                                           \
    do {                                                                             \
        STAMP(b, k);                                                                 \
        if (((b).debug >> 16) == (uint32_t)(k)) return ret;                          \
    } while (0)
#else
#define STAMP_BEGIN() do {} while (0)
#define STAMP(b, k) do {} while (0)
#define STAMP_T(b, k, ret) do {} while (0)
#endif

DEV int lane_id() { return (int)(threadIdx.x & 63); }

/* how many set bits of a 64-lane mask lie below this lane: v_mbcnt_lo + v_mbcnt_hi (no lane mask held in two registers,
 * no 64-bit and + two popcounts) */
DEV int mask_rank(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

DEV uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

template <typename T>
DEV T wave_bcast(T v, int src) {
    return __shfl(v, src, 64);
}

/* THE WAVE HELPERS BELOW NEED ALL 64 LANES ACTIVE.  The ds_bpermute butterflies they replaced (r05) tolerated a partial EXEC
 * mask -- an inactive lane read as 0 --; a DPP scan runs THROUGH the lanes, so inside a lane-divergent branch or the tail of a
 * `for (i = lane; i < n; i += 64)` loop it returns wrong sums without any error.  Every call site is in wave-uniform control
 * flow; a build with -DPYA_CHECK_EXEC (PYA_DEFS=-DPYA_CHECK_EXEC python -m pyascore_amd.build) traps in any helper entered
 * with a partial mask -- the GPU suite is run on such a build once per round (profiles/r06_check_exec.txt).  A caller in
 * divergent flow has to use __shfl_xor / __shfl_up itself (wave_sum_u64, wave_min_f64 below still do). */
#ifdef PYA_CHECK_EXEC
#define PYA_FULL_WAVE() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define PYA_FULL_WAVE() do {} while (0)
#endif

/* Reductions over the 64 lanes by DPP: the inclusive-scan sequence (four shifts inside the rows of 16 lanes, lane 15 of a row
 * broadcast into the next, lane 31 into the upper half), the result read from lane 63 -- 13 instructions where six
 * ds_bpermute exchanges with their address arithmetic were 30.  Every lane must be active (the scan runs through the lanes:
 * every call site is in wave-uniform control flow).  The value returned is wave-uniform. */
#define PYA_DPP_STEP32(x, OP, IDENT, CTRL, ROWS)                                              \
    {                                                                                         \
        const int y_ = __builtin_amdgcn_update_dpp((int)(IDENT), (x), (CTRL), (ROWS), 0xf, false); \
        (x) = OP((x), y_);                                                                    \
    }
#define PYA_DPP_REDUCE32(x, OP, IDENT)            \
    PYA_DPP_STEP32(x, OP, IDENT, 0x111, 0xf)      \
    PYA_DPP_STEP32(x, OP, IDENT, 0x112, 0xf)      \
    PYA_DPP_STEP32(x, OP, IDENT, 0x114, 0xf)      \
    PYA_DPP_STEP32(x, OP, IDENT, 0x118, 0xf)      \
    PYA_DPP_STEP32(x, OP, IDENT, 0x142, 0xa)      \
    PYA_DPP_STEP32(x, OP, IDENT, 0x143, 0xc)
DEV int pya_op_add_i32(int a, int b) { return a + b; }
DEV int pya_op_max_u32(int a, int b) { return (uint32_t)b > (uint32_t)a ? b : a; }
DEV int pya_op_min_u32(int a, int b) { return (uint32_t)b < (uint32_t)a ? b : a; }
DEV int pya_op_max_f32(int a, int b) { return __int_as_float(b) > __int_as_float(a) ? b : a; }
DEV int pya_op_min_f32(int a, int b) { return __int_as_float(b) < __int_as_float(a) ? b : a; }
DEV int wave_sum_i32(int v) {
    PYA_FULL_WAVE();
    int x = v;
    PYA_DPP_REDUCE32(x, pya_op_add_i32, 0)
    return __builtin_amdgcn_readlane(x, 63);
}
DEV uint32_t wave_max_u32(uint32_t v) {
    PYA_FULL_WAVE();
    int x = (int)v;
    PYA_DPP_REDUCE32(x, pya_op_max_u32, 0)
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
DEV uint32_t wave_min_u32(uint32_t v) {
    PYA_FULL_WAVE();
    int x = (int)v;
    PYA_DPP_REDUCE32(x, pya_op_min_u32, -1)
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
DEV float wave_max_f32(float v) {
    PYA_FULL_WAVE();
    int x = __float_as_int(v);
    PYA_DPP_REDUCE32(x, pya_op_max_f32, 0xff800000u)
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
DEV float wave_min_f32(float v) {
    PYA_FULL_WAVE();
    int x = __float_as_int(v);
    PYA_DPP_REDUCE32(x, pya_op_min_f32, 0x7f800000u)
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
DEV double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
DEV double wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
/* exclusive prefix sum over the wave; *total receives the wave sum */
/* Inclusive prefix sum of a 32-bit word over the 64 lanes, or over each half of 32 separately (HALVES): the DPP sequence of
 * wave_excl_scan_i32 below; packed fields add as the integers they are.  Every lane must be active. */
template <bool HALVES>
DEV uint32_t wave_incl_scan_u32(uint32_t v) {
    PYA_FULL_WAVE();
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    if (!HALVES) x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return (uint32_t)x;
}

/* Exclusive prefix sum over the 64 lanes (every lane must be active) and the total.  Six DPP additions -- four shifts inside
 * the rows of 16 lanes, then lane 15 of a row broadcast to the next and lane 31 to the upper half (the sequence LLVM's
 * atomic optimizer builds for gfx9) -- where six ds_bpermute round trips with their address arithmetic were 40 instructions. */
DEV int wave_excl_scan_i32(int v, int *total) {
    PYA_FULL_WAVE();
#ifdef PYA_SCAN_BPERMUTE
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int y = __shfl_up(x, o, 64);
        if (lane_id() >= o) x += y;
    }
#else
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
#endif
    *total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

/* Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Remapping
 * block b to slot (b % 8) * ceil(n / 8) + b / 8 gives every XCD one contiguous range of PSMs, so
 * the per-PSM metadata lines (offsets, counts, status: 8-16 PSMs per 64-byte line) are fetched by
 * one L2 instead of eight. */
DEV uint32_t xcd_slot(uint32_t b, uint32_t n) {
#ifdef PYA_NO_XCD_REMAP
    return b;
#else
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, k = b >> 3;
    return x * q + (x < r ? x : r) + k;                   /* XCD x owns q + (x < r) consecutive slots */
#endif
}

/* LDS traffic of one wave is in order, but the compiler must not move LDS accesses across the
 * points where lanes exchange data through LDS.  The fences name the LDS address space only: a
 * fence over all address spaces makes the compiler drain the vector-memory counter as well
 * (s_waitcnt vmcnt(0)), i.e. every hand-over point would wait for every global load and store in
 * flight -- a full round trip to memory per sync point, and the end of any fetch-ahead. */
DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

/* ---------------------------------------------------------------------------------------
 * Theoretical fragment m/z.  ModifiedPeptide.cpp:570-591: float (running - loss) widened to
 * double, ion-type offsets added one at a time in double, charge applied in double, narrowed.
 * ------------------------------------------------------------------------------------- */
DEV double type_offset(double m, uint8_t type) {
    if (type == 'y') {
        m += 18.010565;
    } else if (type == 'z') {
        m += 18.010565;
        m -= 17.026549;
    } else if (type == 'Z') {
        m += 18.010565;
        m -= 16.018724;
    } else if (type == 'c') {
        m += 17.026549;
    }
    return m;
}
/* ion-type offsets as (m + A) - B in double: b (0,0), c (+NH3,0), y (+H2O,0), z (+H2O,-NH3),
 * Z (+H2O,-NH2); adding or subtracting 0.0 is exact, so this equals ModifiedPeptide.cpp:573-583 */
DEV void type_constants(uint8_t type, double *A, double *B) {
    *A = (type == 'b') ? 0.0 : (type == 'c' ? 17.026549 : 18.010565);
    *B = (type == 'z') ? 17.026549 : (type == 'Z' ? 16.018724 : 0.0);
}

/* the (up to 8) ion-type letters of the configuration as one wave-uniform 64-bit value */
DEV uint64_t load_types64(const DevConfig *cfg) {
    const uint32_t lo = *(const uint32_t *)&cfg->types[0], hi = *(const uint32_t *)&cfg->types[4];
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
DEV uint8_t type_at(uint64_t types64, int t) { return (uint8_t)(types64 >> (8 * t)); }

/* e / d for e*d < 2^32 with a precomputed multiplier (integer division is ~40 instructions) */
struct FastDiv {
    uint32_t d, m;
};
DEV FastDiv fastdiv_make(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1u;
    f.m = (uint32_t)__builtin_ceil(4294967296.0 / (double)f.d);
    return f;
}
DEV uint32_t fastdiv(uint32_t e, const FastDiv &f) {
    if (f.d == 1u) return e;                             /* 2^32 / 1 does not fit the multiplier */
    uint32_t q = __umulhi(e, f.m);
    return q * f.d > e ? q - 1u : q;                     /* the multiplier can be one too large */
}


/* RN(1 / z) for z = 1 .. 255, evaluated by the compiler (IEEE double, round to nearest) */
struct RcpTable {
    double v[256];
    constexpr RcpTable() : v() {
        for (int i = 1; i < 256; i++) v[i] = 1.0 / (double)i;
    }
};
__device__ static const RcpTable kRcpZ = RcpTable();

DEV float charge_mz(double m, int z) {
    if (z == 1) return (float)(m + 1.007825);             /* (m + 1*P)/1 is exact in both steps */
    /* dividing by a power of two is a multiplication by its (exact) reciprocal: same bits as the IEEE
     * division, without the f64 divide sequence */
    if (z == 2) return (float)((m + 2.0 * 1.007825) * 0.5);
    if (z == 4) return (float)((m + 4.0 * 1.007825) * 0.25);
    /* Any other charge: the CORRECTLY ROUNDED quotient a / z without the divide sequence (v_div_scale x 2, v_rcp_f64, two
     * Newton steps, v_div_fmas, v_div_fixup: ~25 quarter-rate instructions where this is 5 fused multiply-adds).  Markstein's
     * theorem (Muller et al., Handbook of Floating-Point Arithmetic, "correctly rounded division with an FMA"): with
     * y = RN(1 / b) and q a faithful rounding of a / b, r = a - b q is exact in an FMA and RN(q + r y) = RN(a / b) (b's
     * significand not all ones: b is an integer below 256).  q0 = RN(a y) is within 1.5 ulp of a / b; one correction makes
     * it faithful (its error is 1/2 ulp + 1.5 ulp x 2^-53), the second correct.  Checked against the division on
     * 494 000 000 random operands for every z up to 255 (no difference; none after the first correction either). */
    /* (charge 3, the common one, from an immediate: lanes with their own charges would fetch the table entry per lane) */
    const double zd = (double)z, a = m + zd * 1.007825;
    double y = 1.0 / 3.0;
    if (z != 3) y = kRcpZ.v[z & 255];
    const double q0 = a * y;
    const double q1 = __builtin_fma(__builtin_fma(-zd, q0, a), y, q0);
    return (float)__builtin_fma(__builtin_fma(-zd, q1, a), y, q1);
}

/* ---------------------------------------------------------------------------------------
 * Match lookup = net semantics of consumePeak/hasMatch/getMatch (ModifiedPeptide.cpp:126-150):
 *   min rank over retained peaks p with f32(f-err) < p < f32(f+err) and f >= p - 0.5.
 * Retained peaks are staged in LDS sorted by m/z (float); a coarse m/z grid gives the start of
 * the scan (the grid only narrows the search, the window test itself is the reference's).
 * ------------------------------------------------------------------------------------- */

struct PeakTable {
    const PeakEntry *e;     /* LDS, ascending m/z, n entries + PYA_TABLE_PAD (+inf, rank 15);     */
                            /* NULL = not staged: look up in the global arrays below             */
    const PeakEntry *g_e;   /* the PSM's retained table in the workspace (ascending m/z, 16-byte aligned) */
    const uint16_t *cell;   /* LDS [PYA_GRID_CELLS]: first peak index whose cell is >= c          */
    const uint16_t *g_cell; /* the same grid in global memory (score_signatures leaves it there)  */
    int n;
    float err;
    float base;             /* m/z of the first retained peak                                    */
    float inv_w;            /* cells per m/z                                                     */
    float nb;               /* -(base * inv_w): cell(x) = fma(x, inv_w, nb), one instruction      */
    int last_cell;
    bool half_check;        /* mz_error > 0.49: the reference's lower_bound(mz - .5) can bite    */
};

/* LDS bytes of a staged table for up to `cap` retained peaks */
DEV size_t peak_table_bytes(uint32_t cap) { return ((size_t)cap + PYA_TABLE_PAD) * sizeof(PeakEntry); }

/* Copies R retained peaks (workspace, 16-byte aligned) into an LDS table (16-byte aligned) and appends the
 * PYA_TABLE_PAD sentinels: two entries per lane and load, every load issued before the first store.  `nt`
 * lanes of `tid` 0 .. nt-1 share the work (64 for a wavefront, the workgroup's size for score_big). */
DEV void copy_peak_table(const PeakEntry *src, int R, PeakEntry *dst, int tid, int nt) {
    const uint4 *s4 = (const uint4 *)src;
    uint4 *d4 = (uint4 *)dst;
    const int pairs = (R + 1) >> 1;
    const uint32_t inf = __float_as_uint(__builtin_huge_valf());
    for (int base = 0; base < pairs; base += 4 * nt) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = base + u * nt + tid;
            if (q < pairs) v[u] = s4[q];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int q = base + u * nt + tid;
            if (q < pairs) {
                if (2 * q + 1 >= R) {                        /* the odd tail: the first sentinel */
                    v[u].z = inf;
                    v[u].w = PYA_NO_MATCH;
                }
                d4[q] = v[u];
            }
        }
    }
    const int first_pad = (R + 1) & ~1;                      /* sentinels from there to R + PYA_TABLE_PAD - 1 */
    if (tid < PYA_TABLE_PAD) {
        const int at = first_pad + tid;
        if (at < R + PYA_TABLE_PAD) {
            PeakEntry x;
            x.mz = __builtin_huge_valf();
            x.rank = PYA_NO_MATCH;
            dst[at] = x;
        }
    }
}

/* copies the retained peaks of `psm` (written by bin_spectra) into LDS and appends sentinels */
DEV void stage_peak_table_at(const BatchDev &b, int64_t p0, int R, PeakEntry *dst, PeakTable *t) {
    copy_peak_table(b.ret + p0, R, dst, lane_id(), 64);
    t->e = dst;
    t->g_cell = nullptr;
    t->g_e = b.ret + p0;
    t->n = R;
    t->err = b.cfg->mz_error;
    t->half_check = b.cfg->mz_error > 0.49f;
}
DEV void stage_peak_table(const BatchDev &b, uint32_t psm, PeakEntry *dst, PeakTable *t) {
    const int64_t p0 = b.ret_off[psm];
    const int R = (int)b.ret_n[psm];
    copy_peak_table(b.ret + p0, R, dst, lane_id(), 64);
    t->e = dst;
    t->g_cell = nullptr;
    t->g_e = b.ret + p0;
    t->n = R;
    t->err = b.cfg->mz_error;
    t->half_check = b.cfg->mz_error > 0.49f;
}

/* monotone non-decreasing in x: float subtract, multiply by a positive constant, truncate */
DEV int grid_cell(const PeakTable &t, float x) {
    /* any function that does not decrease with x serves (a peak above the window's lower bound must not
     * land in an earlier cell than the bound), as long as peaks and bounds go through the same one */
    const float rel = __builtin_fmaf(x, t.inv_w, t.nb);
    /* v_cvt_u32_f32 saturates: below zero (and NaN) to 0, above the range to 2^32 - 1 -- the clamp at
     * zero without an instruction of its own */
    uint32_t c;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(c) : "v"(rel));
    const uint32_t last = (uint32_t)t.last_cell;
    return (int)(c > last ? last : c);
}

/* cell geometry from the first and last retained m/z (same arithmetic wherever it is needed) */
DEV void grid_params(PeakTable *t, float first, float last) {
    t->base = first;
    const float range = last - first;
    float inv_w = 0.125f;                                 /* 8 m/z per cell ...                */
    if (range * inv_w > (float)(PYA_GRID_CELLS - 2)) inv_w = (float)(PYA_GRID_CELLS - 2) / range;
    t->inv_w = inv_w;                                     /* ... or wider to fit the grid      */
    t->nb = -(first * inv_w);
    t->last_cell = PYA_GRID_CELLS - 1;
    t->last_cell = grid_cell(*t, last);                   /* the last peak's own cell: every cell up to it gets filled */
}

/* builds the grid for the n staged peaks (wave-cooperative; caller syncs LDS before and after):
 * cell k = the (even) index at or before the first peak whose cell is >= k.  The first peak of every
 * occupied cell writes its index; the empty cells then take the value of the next occupied one --
 * four cells per lane, the lane's own suffix first, the rest from the nearest lane to the right that
 * has an occupied cell (one ballot, one shuffle).  (A loop that lets every peak fill the cells back to
 * its predecessor's runs as long as the widest gap: 200 vector instructions where this takes 50.) */
DEV void grid_build(PeakTable *t, uint16_t *cell_lds) {
    static_assert(PYA_GRID_CELLS == 256, "four cells per lane");
    const int lane = lane_id();
    t->cell = cell_lds;
    if (t->n <= 0) {
        t->base = 0.f;
        t->inv_w = 0.f;
        t->nb = 0.f;
        t->last_cell = 0;
        if (lane == 0) cell_lds[0] = 0;
        return;
    }
    grid_params(t, t->e[0].mz, t->e[t->n - 1].mz);
    uint64_t *cells4 = (uint64_t *)cell_lds;
    cells4[lane] = ~0ull;
    wave_lds_sync();
    int before = -1;                                      /* cell of the peak before this chunk */
    for (int base = 0; base < t->n; base += 64) {
        const int i = base + lane;
        const bool in = i < t->n;
        const int c = in ? grid_cell(*t, t->e[i].mz) : 0x7fffffff;
        int cp = __shfl_up(c, 1, 64);
        if (lane == 0) cp = before;
        if (in && cp < c) cell_lds[c] = (uint16_t)(i & ~1);   /* even: entries4 reads 16-byte pairs */
        before = __builtin_amdgcn_readlane(c, 63);
    }
    wave_lds_sync();
    const uint64_t w = cells4[lane];
    const uint32_t E = 0xffffu;
    uint32_t v0 = (uint32_t)w & E, v1 = (uint32_t)(w >> 16) & E, v2 = (uint32_t)(w >> 32) & E, v3 = (uint32_t)(w >> 48);
    const uint32_t mine = v0 != E ? v0 : (v1 != E ? v1 : (v2 != E ? v2 : v3));   /* first occupied cell of the lane */
    const uint64_t occupied = __ballot(mine != E);
    const uint64_t right = lane == 63 ? 0ull : occupied & (~0ull << (lane + 1));
    const int src = right ? __builtin_ctzll(right) : lane;
    const uint32_t next = (uint32_t)__shfl((int)mine, src, 64);   /* (none: only past the last peak's cell, never read) */
    v3 = v3 != E ? v3 : next;
    v2 = v2 != E ? v2 : v3;
    v1 = v1 != E ? v1 : v2;
    v0 = v0 != E ? v0 : v1;
    cells4[lane] = (uint64_t)(v0 | (v1 << 16)) | ((uint64_t)(v2 | (v3 << 16)) << 32);
}

/* Branch-light lookup: the grid gives an index at or before the first peak > lo; four entries
 * are fetched at once (the sentinels make this safe) and reduced with selects; only when the
 * window is not closed by the fourth entry does a lane continue with the scalar scan. */
/* the table of `psm` left in global memory (for kernels that make only a handful of lookups) */
DEV void global_peak_table_at(const BatchDev &b, uint32_t psm, int64_t p0, int R, PeakTable *t);
DEV void global_peak_table(const BatchDev &b, uint32_t psm, PeakTable *t) {
    global_peak_table_at(b, psm, b.ret_off[psm], (int)b.ret_n[psm], t);
}
/* (offset and count already known: the packed descriptor, ret_n fetched beside it) */
DEV void global_peak_table_at(const BatchDev &b, uint32_t psm, int64_t p0, int R, PeakTable *t) {
    t->e = nullptr;
    t->cell = nullptr;
    t->g_cell = b.grid + (size_t)psm * PYA_GRID_CELLS;
    t->g_e = b.ret + p0;
    t->n = R;
    t->err = b.cfg->mz_error;
    t->half_check = b.cfg->mz_error > 0.49f;
    t->base = 0.f;
    t->inv_w = 0.f;
    t->nb = 0.f;
    t->last_cell = 0;
    if (t->n > 0) grid_params(t, t->g_e[0].mz, t->g_e[t->n - 1].mz);
}

/* Same window test on the table in the workspace.  The grid score_signatures built for this PSM gives the
 * (even) start; the next four entries come as two 16-byte loads, so a lookup is two dependent round trips to
 * memory instead of the ten of a binary search. */
DEV int match_rank_global(const PeakTable &t, float f) {
    if (t.n <= 0) return PYA_NO_MATCH;
    const float lo = f - t.err;
    const float hi = f + t.err;
    int idx = (int)t.g_cell[grid_cell(t, lo)];            /* every peak > lo has index >= idx (even)  */
    const int last = t.n - 1;
    /* reading up to three entries past the PSM's table stays inside the plan's arena and is masked below */
    const uint4 *p4 = (const uint4 *)(t.g_e + idx);
    const uint4 a = p4[0], c = p4[1];
    const float m[4] = {__uint_as_float(a.x), __uint_as_float(a.z), __uint_as_float(c.x), __uint_as_float(c.z)};
    const int r[4] = {(int)a.y, (int)a.w, (int)c.y, (int)c.w};
    int best = PYA_NO_MATCH;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bool in = idx + j <= last && m[j] > lo && m[j] < hi;
        if (t.half_check) in = in && (double)f >= (double)m[j] - 0.5;
        best = in && r[j] < best ? r[j] : best;
    }
    if (idx + 3 < last && m[3] < hi) {                    /* rare: more than four entries to look at */
        for (idx += 4; idx <= last; idx++) {
            const PeakEntry x = t.g_e[idx];
            if (!(x.mz < hi)) break;
            if (x.mz > lo && (!t.half_check || (double)f >= (double)x.mz - 0.5)) {
                const int rr = (int)x.rank;
                best = rr < best ? rr : best;
            }
        }
    }
    return best;
}

/* Four entries from the even index at or below `idx`, as two 16-byte reads: the LDS serves a
 * ds_read_b128 in 4 cycles and a ds_read2_b64 (what four 8-byte entries from an odd index become) in
 * 8, and the lookups of a walk are what fills the LDS pipe.  An entry below idx sits in an earlier
 * grid cell than the window's lower bound, so it fails the window test by itself.  (t.e is 16-byte
 * aligned in every kernel's LDS layout.)  Returns the index after the four. */
DEV int entries4(const PeakTable &t, uint32_t idx, PeakEntry *e) {
    const uint32_t base = idx;                               /* (grid_build stores even indices) */
    const uint4 *p = (const uint4 *)__builtin_assume_aligned((const unsigned char *)t.e + base * 8u, 16);
    const uint4 a = p[0], c = p[1];
    e[0].mz = __uint_as_float(a.x); e[0].rank = a.y;
    e[1].mz = __uint_as_float(a.z); e[1].rank = a.w;
    e[2].mz = __uint_as_float(c.x); e[2].rank = c.y;
    e[3].mz = __uint_as_float(c.z); e[3].rank = c.w;
    return (int)base + 4;
}

DEV int match_rank_lds(const PeakTable &t, float f) {
    const float lo = f - t.err;
    const float hi = f + t.err;
    int idx = (int)t.cell[grid_cell(t, lo)];              /* every peak > lo has index >= idx  */
    PeakEntry e4[4];
    const int next = entries4(t, (uint32_t)idx, e4);
    const PeakEntry e0 = e4[0], e1 = e4[1], e2 = e4[2], e3 = e4[3];
    int best = PYA_NO_MATCH;
    if (!t.half_check) {
        int r;
        r = e0.mz > lo ? (int)e0.rank : PYA_NO_MATCH; r = e0.mz < hi ? r : PYA_NO_MATCH; best = r < best ? r : best;
        r = e1.mz > lo ? (int)e1.rank : PYA_NO_MATCH; r = e1.mz < hi ? r : PYA_NO_MATCH; best = r < best ? r : best;
        r = e2.mz > lo ? (int)e2.rank : PYA_NO_MATCH; r = e2.mz < hi ? r : PYA_NO_MATCH; best = r < best ? r : best;
        r = e3.mz > lo ? (int)e3.rank : PYA_NO_MATCH; r = e3.mz < hi ? r : PYA_NO_MATCH; best = r < best ? r : best;
        if (e3.mz < hi) {                                  /* rare: more than four entries to look at */
            idx = next;
            for (;;) {
                const PeakEntry x = t.e[idx];
                if (!(x.mz < hi)) break;
                if (x.mz > lo) best = (int)x.rank < best ? (int)x.rank : best;
                idx++;
            }
        }
    } else {
        for (;;) {
            const PeakEntry x = t.e[idx];
            if (!(x.mz < hi)) break;
            if (x.mz > lo && (double)f >= (double)x.mz - 0.5) best = (int)x.rank < best ? (int)x.rank : best;
            idx++;
        }
    }
    return best;
}

/* The same lookup in two parts, so that a caller can have several lookups in flight: look4 is
 * straight-line (grid cell, four entries, select-reduce) and says whether the window extends past
 * the fourth entry; look_rest finishes those (rare).  Only for mz_error <= 0.49 (no half_check). */
struct Look {
    int best, idx;
    float lo, hi, last;       /* last: m/z of the fourth entry */
    /* the window extends past the fourth entry (rare).  Callers branch on it per lane -- the compiler
     * skips the block when no lane needs it, and a wave-wide "any" would cost two vector instructions */
    DEV bool more() const { return last < hi; }
};
DEV Look look4(const PeakTable &t, float f) {
    Look k;
    k.lo = f - t.err;
    k.hi = f + t.err;
    PeakEntry e4[4];
    k.idx = entries4(t, (uint32_t)t.cell[grid_cell(t, k.lo)], e4); /* (where look_rest resumes) */
    const PeakEntry e0 = e4[0], e1 = e4[1], e2 = e4[2], e3 = e4[3];
    /* one select per entry, then min3 + min (ranks are at most PYA_NO_MATCH = 15) */
    const int r0 = (e0.mz > k.lo && e0.mz < k.hi) ? (int)e0.rank : PYA_NO_MATCH;
    const int r1 = (e1.mz > k.lo && e1.mz < k.hi) ? (int)e1.rank : PYA_NO_MATCH;
    const int r2 = (e2.mz > k.lo && e2.mz < k.hi) ? (int)e2.rank : PYA_NO_MATCH;
    const int r3 = (e3.mz > k.lo && e3.mz < k.hi) ? (int)e3.rank : PYA_NO_MATCH;
    const int m01 = r0 < r1 ? r0 : r1, m23 = r2 < r3 ? r2 : r3;
    const int best = m01 < m23 ? m01 : m23;
    k.best = best;
    k.last = e3.mz;
    return k;
}
DEV int look_rest(const PeakTable &t, const Look &k) {
    int best = k.best;
    for (int idx = k.idx;; idx++) {
        const PeakEntry x = t.e[idx];
        if (!(x.mz < k.hi)) break;
        if (x.mz > k.lo) best = (int)x.rank < best ? (int)x.rank : best;
    }
    return best;
}

DEV int match_rank(const PeakTable &t, float f) {
    return t.e ? match_rank_lds(t, f) : match_rank_global(t, f);
}

/* Row n of the score table starts at 10 * (1 + 2 + ... + n) floats: rows are dense from n = 0 and
 * hold PYA_NTOP x (n + 1) entries (score_table.cpp; the host checks this when it uploads). */
DEV uint32_t lut_row(uint32_t n) { return 5u * n * (n + 1u); }

/* rank histogram: 10 x 16-bit fields in three 64-bit words (ranks 0-3 | 4-7 | 8-9) */
struct Hist {
    uint64_t a, b, c;
};
DEV void hist_add(Hist &h, int rank) {
    uint64_t inc = 1ull << ((rank & 3) * 16);
    h.a += rank < 4 ? inc : 0ull;
    h.b += (rank >= 4 && rank < 8) ? inc : 0ull;
    h.c += (rank >= 8 && rank < PYA_NTOP) ? inc : 0ull;
}
DEV uint32_t hist_get(const Hist &h, int r) {
    uint64_t w = r < 4 ? h.a : (r < 8 ? h.b : h.c);
    return (uint32_t)(w >> ((r & 3) * 16)) & 0xffffu;
}
DEV Hist hist_wave_sum(Hist h) {
    h.a = wave_sum_u64(h.a);
    h.b = wave_sum_u64(h.b);
    h.c = wave_sum_u64(h.c);
    return h;
}

/* ---------------------------------------------------------------------------------------
 * Per-residue data of one peptide, one residue per lane (registers, read with v_readlane).
 * ModifiedPeptide.cpp:24-79.
 * ------------------------------------------------------------------------------------- */
struct Residues {
    float m0, m1;           /* unmodified / modified mass of the lane's residue             */
    uint32_t nl;            /* NL class unmodified | modified << 4                          */
    uint64_t site_mask;     /* wave-uniform: bit i = residue i modifiable                   */
    int L;
};

DEV Residues load_residues(const BatchDev &b, const DevConfig *cfg, int64_t psm) {
    Residues r;
    int64_t p0 = b.pep_off[psm];
    r.L = (int)(b.pep_off[psm + 1] - p0);
    int i = lane_id();
    bool in = i < r.L;
    uint32_t c = in ? (uint32_t)b.pep[p0 + i] : (uint32_t)'A';
    uint32_t li = (c - 'A') & 31u;
    float m0 = cfg->res_mass[li];
    bool modifiable = in && (cfg->res_modifiable[li] || (cfg->allow_n && i == 0) ||
                             (cfg->allow_c && i == r.L - 1));
    float m1 = m0 + cfg->mod_mass;
    uint32_t nl0 = cfg->nl_upper[li];
    uint32_t nl1 = modifiable ? (uint32_t)cfg->nl_lower[li] : 0u;
    /* fixed modifications (ModifiedPeptide.cpp:59-79): fetched 64 at a time, one per lane, so the
     * loop over them runs on registers instead of one dependent memory round trip per entry */
    const int64_t a0 = b.aux_off[psm], a1 = b.aux_off[psm + 1];
    for (int64_t base = a0; base < a1; base += 64) {
        const int n = (int)(a1 - base < 64 ? a1 - base : 64);
        uint32_t my_pos = 0;
        float my_am = 0.f;
        if (i < n) {
            my_pos = b.aux_pos[base + i];
            my_am = b.aux_mass[base + i];
        }
        for (int j = 0; j < n; j++) {
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)my_pos, j);
            const float am = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_am), j));
            const int idx = pos > 0 ? (int)pos - 1 : 0;
            if (idx == i) {
                m0 += am;
                m1 += am;
                if (cfg->nl_lower[li]) nl0 = cfg->nl_lower[li];
            }
        }
    }
    r.m0 = m0;
    r.m1 = m1;
    r.nl = nl0 | (nl1 << 4);
    r.site_mask = __ballot(modifiable);
    return r;
}

/* r06 -- the same from the PSM's packed descriptor (BatchDev.desc: one cache line, common.h) and with the per-letter tables
 * of the configuration taken into registers up front: a PSM's prologue is then TWO rounds of loads -- {descriptor, status,
 * the 32 table entries, one per lane} and {letters, fixed modifications, whatever the caller fetches at the descriptor's
 * offsets} -- where load_residues behind the callers' status / count tests made five or six, one after the other (offsets,
 * letters, a gather from the tables by letter: r06 stamps, a fifth of score_big's and of the lean localize kernel's wave
 * time went by before the first useful instruction). */
struct PsmDesc {
    int64_t ret0, pep0, sig0, aux0;     /* offsets of the retained table, the letters, the site assignments' arrays, the fixed modifications */
    int L, n_aux, k, n_sites, zmax;
    uint32_t N, order_off;
};
DEV PsmDesc load_desc(const BatchDev &b, uint32_t psm) {
    const uint64_t *dw = b.desc + (size_t)psm * PYA_DESC_WORDS;
    const uint64_t w4 = dw[4], w5 = dw[5];
    PsmDesc d;
    d.ret0 = (int64_t)dw[0];
    d.pep0 = (int64_t)dw[1];
    d.sig0 = (int64_t)dw[2];
    d.aux0 = (int64_t)dw[3];
    d.L = (int)(w4 & 0xffffu);
    d.n_aux = (int)((w4 >> 16) & 0xffffu);
    d.k = (int)((w4 >> 32) & 0xffffu);
    d.n_sites = (int)((w4 >> 48) & 0xffu);
    d.zmax = (int)(w4 >> 56);
    d.N = (uint32_t)w5;
    d.order_off = (uint32_t)(w5 >> 32);
    return d;
}
/* the configuration's per-letter tables, entry (lane & 31) in every lane: issue before anything is waited for */
struct LetterRegs {
    float mass;
    uint32_t flags;                     /* modifiable | nl_upper << 8 | nl_lower << 16 */
};
DEV LetterRegs load_letter_regs(const DevConfig *cfg) {
    const int l = lane_id() & 31;
    LetterRegs t;
    t.mass = cfg->res_mass[l];
    t.flags = (uint32_t)cfg->res_modifiable[l] | ((uint32_t)cfg->nl_upper[l] << 8) | ((uint32_t)cfg->nl_lower[l] << 16);
    return t;
}
DEV Residues load_residues_desc(const BatchDev &b, const DevConfig *cfg, const PsmDesc &d, const LetterRegs &t) {
    Residues r;
    r.L = d.L;
    const int i = lane_id();
    const bool in = i < r.L;
    const uint32_t c = in ? (uint32_t)b.pep[d.pep0 + i] : (uint32_t)'A';
    const uint32_t li = (c - 'A') & 31u;
    float m0 = __shfl(t.mass, (int)li, 64);                       /* (lanes 0 .. 31 hold the table) */
    const uint32_t fl = (uint32_t)__shfl((int)t.flags, (int)li, 64);
    const uint32_t nl_lower = (fl >> 16) & 0xffu;
    const bool modifiable = in && ((fl & 0xffu) || (cfg->allow_n && i == 0) || (cfg->allow_c && i == r.L - 1));
    float m1 = m0 + cfg->mod_mass;
    uint32_t nl0 = (fl >> 8) & 0xffu;
    const uint32_t nl1 = modifiable ? nl_lower : 0u;
    /* fixed modifications (ModifiedPeptide.cpp:59-79), as load_residues applies them */
    for (int base = 0; base < d.n_aux; base += 64) {
        const int n = d.n_aux - base < 64 ? d.n_aux - base : 64;
        uint32_t my_pos = 0;
        float my_am = 0.f;
        if (i < n) {
            my_pos = b.aux_pos[d.aux0 + base + i];
            my_am = b.aux_mass[d.aux0 + base + i];
        }
        for (int j = 0; j < n; j++) {
            const uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)my_pos, j);
            const float am = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_am), j));
            const int idx = pos > 0 ? (int)pos - 1 : 0;
            if (idx == i) {
                m0 += am;
                m1 += am;
                if (nl_lower) nl0 = nl_lower;
            }
        }
    }
    r.m0 = m0;
    r.m1 = m1;
    r.nl = nl0 | (nl1 << 4);
    r.site_mask = __ballot(modifiable);
    return r;
}

/* sig bits (bit j = j-th modifiable residue) -> residue mask (bit i = residue i modified).
 * `site_mask` is wave-uniform everywhere this is called, so the loop runs over the sites on the
 * scalar unit (s_ff1 / s_bitset) and each lane spends a handful of VALU instructions per site,
 * where a per-lane "while (bits)" costs a divergent loop of 64-bit updates. */
DEV uint64_t deposit_sites(uint64_t bits, uint64_t site_mask) {
    uint64_t out = 0, m = site_mask;
    for (int j = 0; m; j++) {
        const int pos = __builtin_ctzll(m);
        m &= m - 1;
        out |= ((bits >> j) & 1ull) << pos;
    }
    return out;
}

/* position of the n-th (0-based) set bit of the wave-uniform mask m (n may differ per lane) */
DEV int nth_set_bit(uint64_t m, int n) {
    int res = 64;
    for (int j = 0; m; j++) {
        const int pos = __builtin_ctzll(m);
        m &= m - 1;
        res = j == n ? pos : res;
    }
    return res;
}

DEV uint32_t nl_bump(uint32_t state, uint32_t cls) {
    /* 2 bits per class, saturating at 2; cls in 1..4 (the fast kernels: an 8-bit state), 1..8 in the general kernel */
    uint32_t sh = (cls - 1) * 2;
    uint32_t cur = (state >> sh) & 3u;
    return cur < 2u ? state + (1u << sh) : state;
}

#endif
