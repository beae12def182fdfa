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
#include "localize_core.hip.h"

extern "C" size_t pya_localize_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap,
                                         uint32_t pool_cap, uint32_t sb) {
    /* (the localize kernel looks peaks up in global memory: no peak table here) */
    size_t fixed = 512 + PYA_MAX_UNIQ * 4 + (size_t)push_cap * 16 + 64 * 16 + 16;
    size_t srt = (size_t)n_cap * 10 + 64;
    size_t lst = pya_loc_lds_bytes(pos_cap, pool_cap, sb);
    return fixed + (srt > lst ? srt : lst) + 64;
}

#ifndef LOC_WAVES
#define LOC_WAVES 4            /* general instantiation: 128 VGPRs */
#endif
#ifndef LOC_WAVES_PLAIN
#define LOC_WAVES_PLAIN 5      /* lean instantiation: fits 102 VGPRs without scratch */
#endif

/* One PSM, one wavefront.  Two instantiations:
 *   PLAIN = true  -- no neutral losses, fragment charge 1, every residue mass positive (so every
 *                    fragment list is ascending and position-indexed), summary mode.  Everything
 *                    only other PSMs need is compiled out, which takes the kernel from 128 to
 *                    under 102 VGPRs = 5 instead of 4 waves per SIMD (this kernel's speed follows
 *                    its occupancy).  Returns true -- nothing written -- when the PSM turns out to
 *                    need a route it does not have: an ion with two partners within mz_error,
 *                    introsort running out of depth, a non-positive residue mass.  Such PSMs are appended to a list and redone by
 *                    the general instantiation.
 *   PLAIN = false -- everything. */
template <bool PLAIN>
DEV bool localize_body(const BatchDev &b, uint32_t psm, unsigned char *lds_raw, uint32_t push_cap, uint32_t pos_cap,
                       uint32_t pool_cap, uint32_t sb, uint32_t gtp) {
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
        return false;
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
        return false;
    }

    STAMP_BEGIN();
    /* only a handful of ions are matched here, so the retained-peak table is not staged in LDS:
     * that keeps this kernel's LDS small (occupancy) and saves the staging + grid build */
    K3Lds lds = carve(lds_raw, 0, false, push_cap);
    LocCtx ctx;
    ctx.b = &b;
    ctx.cfg = cfg;
    stage_tables(b, cfg, lds, psm, &ctx.tab, &ctx.nl, false);
    const Residues res = load_residues(b, cfg, psm);
    const uint64_t site_mask_u = res.site_mask;
    /* positive residue masses make the float32 running sum, hence every m/z list, ascending */
    const int zmax = b.max_charge[psm];
    const bool presorted = zmax == 1 && cfg->n_nl == 0 &&
                           !__any(lane < res.L && !(res.m0 > 0.f && res.m1 > 0.f));
    if (PLAIN && (!presorted || b.keep || (b.debug & 512))) return true;   /* not this kernel's PSM */
    STAMP(b, 20);

    /* ---- sort (cpp/Ascore.cpp:141-146) ---- */
    if (lane == 0) *lds.n_pushed = 0;
    lds.site_max[lane] = 0;
    lds.site_tie[lane] = 0;
    lds.site_alt[lane] = 0ull;
    /* The winner is the front of the sorted list: the largest PepScore, and among equal ones
     * whichever std::sort leaves first.  When the maximum is unique (4 PSMs in 5) no emulation is
     * needed to name it. */
    const float ws_lane = lane < N ? ws[lane] : 0.f;       /* the first 64 scores stay in a register */
    uint32_t kmax = 0;
    for (int i = lane; i < N; i += 64) {
        const uint32_t u = __float_as_uint(i < 64 ? ws_lane : ws[i]);   /* scores are >= 0: bit order = value order */
        kmax = u > kmax ? u : kmax;
    }
    kmax = wave_max_u32(kmax);
    int n_max = 0;
    uint32_t first_max = 0xffffffffu;
    for (int i = lane; i < N; i += 64) {
        if (__float_as_uint(i < 64 ? ws_lane : ws[i]) == kmax) {
            n_max++;
            first_max = first_max < (uint32_t)i ? first_max : (uint32_t)i;
        }
    }
    n_max = wave_sum_i32(n_max);
    first_max = wave_min_u32(first_max);
    uint32_t best_i = first_max;
    STAMP(b, 21);
    if (n_max != 1 || b.keep || (b.debug & 1024)) {
        SortLds srt;
        srt.key = (float *)lds.scratch;
        srt.idx = (uint16_t *)(srt.key + N);
        srt.lpos = srt.idx + N;
        srt.rpos = srt.lpos + N;
        for (int i = lane; i < N; i += 64) {
            srt.key[i] = i < 64 ? ws_lane : ws[i];
            srt.idx[i] = (uint16_t)i;
        }
        wave_lds_sync();
        /* only the left spine of the partition tree decides the front element; the full sort is
         * needed when the caller wants the whole ordering */
        if (!(b.debug & 8)) {
            if (sort_introsort_loop<PLAIN>(srt, N, b.keep == 0)) return true;
        }
        /* front of the sorted list = left-most maximum of the partitioned array */
        uint32_t first_pos = 0xffffffffu;
        for (int i = lane; i < N; i += 64)
            if (__float_as_uint(srt.key[i]) == kmax) first_pos = first_pos < (uint32_t)i ? first_pos : (uint32_t)i;
        first_pos = wave_min_u32(first_pos);
        best_i = srt.idx[first_pos];
        if (b.keep) {
            for (int i = lane; i < N; i += 64) b.sorted_idx[s0 + sort_final_pos(srt, i, N)] = srt.idx[i];
        }
    }
    STAMP(b, 22);
    const float best_ws = __uint_as_float(kmax);
    const uint64_t best_bits = order[best_i];
    wave_lds_sync();

    STAMP(b, 23);
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
                        if ((double)__builtin_fabsf(best_ws - __uint_as_float(u)) < 1e-6) {
                            /* ties the winner: Ascore 0 (Ascore.cpp:159-161), no ion work needed */
                            lds.site_tie[a] = 1u;
                            atomicOr(&lds.site_alt[a], 1ull << nth_set_bit(site_mask_u, __builtin_ctzll(came)));
                            continue;
                        }
                        const uint32_t slot = atomicAdd(lds.n_pushed, 1u);
                        if (slot < push_cap) {
                            PushedEntry pe;
                            pe.bits = c;
                            pe.ws = __uint_as_float(u);
                            pe.idx = (uint32_t)i;
                            lds.pushed[slot] = pe;
                        }
                    }
                }
            }
        }
        wave_lds_sync();
    }
    const uint32_t n_pushed = *lds.n_pushed;
    int fail = 0;
    if (n_pushed > push_cap) fail = 2;                    /* cannot happen: push_cap >= k * (n_sites - k) */
    uint32_t np = n_pushed < push_cap ? n_pushed : push_cap;
    if (b.debug & 16) np = 0;

    STAMP(b, 24);
    /* ---- Ascores, sb-1 competitors at a time ---- */
    ctx.w = loc_carve(lds.scratch, pos_cap, pool_cap, sb);
    ctx.sb = (int)sb;
    ctx.gtp = (int)gtp;
    ctx.L = res.L;
    ctx.zmax = zmax;
    ctx.presorted = presorted;
    ctx.pos_cap = pos_cap;
    ctx.pool_cap = pool_cap;
    const LocLds &w = ctx.w;
    w.m0[lane] = res.m0;
    w.m1[lane] = res.m1;
    w.nlp[lane] = (uint8_t)res.nl;
    if (lane == 0) w.sig_mask[0] = deposit_sites(best_bits, res.site_mask);
    wave_lds_sync();

    STAMP(b, 25);
    float my_asc = __builtin_huge_valf();     /* lane a keeps site a */
    uint64_t my_alt = 0ull;
    const bool declined = loc_ascore_all<PLAIN>(ctx, lds.pushed, np, lds.site_alt, b.rec + s0 * PYA_REC_WORDS,
                   best_bits, best_ws, best_i, res.site_mask,
                   &my_asc, &my_alt, &fail);
    if (PLAIN && declined) return true;
    STAMP(b, 36);
    if (lane < k && lds.site_tie[lane]) my_asc = 0.f < my_asc ? 0.f : my_asc;
    if (lane < k) my_alt |= lds.site_alt[lane];
    if (lane < k && lane < (int)max_k) {
        out_asc[lane] = my_asc;
        out_alt[lane] = my_alt;
    }
    const bool any_fail = __any(fail != 0), overflow = __any(fail == 2);
    if (lane == 0) {
        b.best_score[psm] = best_ws;
        b.best_sig[psm] = best_bits;
        b.n_sig_out[psm] = N;
        if (any_fail) b.status[psm] = overflow ? PYA_ST_PUSHED_OVERFLOW : PYA_ST_LUT_RANGE;
    }
    return false;
}

template <bool PLAIN>
__global__ __launch_bounds__(64, PLAIN ? LOC_WAVES_PLAIN : LOC_WAVES) void pya_localize_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap,
    uint32_t sb, uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const bool declined = localize_body<PLAIN>(b, psm, lds_raw, push_cap, pos_cap, pool_cap, sb, gtp);
    if (PLAIN && declined && lane_id() == 0) b.redo3_ids[atomicAdd(b.redo3_count, 1u)] = psm;
}

/* the PSMs the lean instantiation declined, on the general one: a small grid strides over the list */
__global__ __launch_bounds__(64, LOC_WAVES) void pya_localize_redo_kernel(BatchDev b, uint32_t push_cap,
                                                                        uint32_t pos_cap, uint32_t pool_cap,
                                                                        uint32_t sb, uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t n = *b.redo3_count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        localize_body<false>(b, b.redo3_ids[k], lds_raw, push_cap, pos_cap, pool_cap, sb, gtp);
        wave_lds_sync();
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
    sort_introsort_loop<false>(srt, N, false);
    for (int i = lane; i < N; i += 64) perm[sort_final_pos(srt, i, N)] = srt.idx[i];
}

extern "C" int pya_launch_localize(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids,
                                   uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap,
                                   uint32_t sb, uint32_t gtp, uint32_t plain, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = pya_localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
    hipError_t e;
    if (!plain) {
        e = hipFuncSetAttribute((const void *)pya_localize_kernel<false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(pya_localize_kernel<false>, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, push_cap,
                           pos_cap, pool_cap, sb, gtp);
        return (int)hipGetLastError();
    }
    /* lean instantiation first, then whatever it declined on the general one */
    e = hipMemsetAsync(b->redo3_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)pya_localize_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_localize_kernel<true>, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, push_cap,
                       pos_cap, pool_cap, sb, gtp);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void *)pya_localize_redo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = n_ids < 8192u ? n_ids : 8192u;
    hipLaunchKernelGGL(pya_localize_redo_kernel, dim3(grid), dim3(64), lds, stream, *b, push_cap, pos_cap, pool_cap,
                       sb, gtp);
    return (int)hipGetLastError();
}

extern "C" size_t pya_amb_lds_bytes(uint32_t peak_cap, uint32_t list_cap) {
    return 512 + PYA_MAX_UNIQ * 4 + PYA_MAX_PUSHED * 16 + 64 * 16 + 16 + PYA_GRID_CELLS * 2 +
           ((size_t)peak_cap + PYA_TABLE_PAD) * 8 + (size_t)list_cap * 10 + 128;
}

extern "C" int pya_launch_ambiguity(const BatchDev *b, uint32_t psm, uint32_t peak_cap, uint32_t list_cap,
                                    uint64_t ref_bits, uint64_t oth_bits, const float *d_scores,
                                    float ref_ws, float oth_ws, float *d_out, hipStream_t stream) {
    size_t lds = pya_amb_lds_bytes(peak_cap, list_cap);
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
