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
#include "localize_body.hip.h"

extern "C" size_t pya_localize_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap,
                                         uint32_t pool_cap, uint32_t sb) {
    return localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
}

template <bool PLAIN>
__global__ __launch_bounds__(64, PLAIN ? LOC_WAVES_PLAIN : LOC_WAVES) void pya_localize_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap,
    uint32_t sb, uint32_t gtp, uint32_t sort_room) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const bool declined = localize_body<PLAIN>(b, psm, lds_raw, push_cap, pos_cap, pool_cap, sb, gtp, sort_room != 0);
    if (PLAIN && declined && lane_id() == 0) b.redo3_ids[atomicAdd(b.redo3_count, 1u)] = psm;
}

/* The general route with the hash-grid search for site-determining ions (localize_hash.hip.h); what it declines goes
 * to the hand-over list and from there to the list-based general instantiation. */
__global__ __launch_bounds__(64, LOC_WAVES_HASH) void pya_localize_hash_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t push_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc,
    uint32_t hs, uint32_t pp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const bool declined = localize_body<false, true>(b, psm, lds_raw, push_cap, pos_cap, (uint32_t)loc_hash_words(vc, hs, pp, sb), sb,
                                                     0u, true, InlineSrc(), vc, hs, pp);
    if (declined && lane_id() == 0) b.redo3_ids[atomicAdd(b.redo3_count, 1u)] = psm;
}

/* the PSMs a lean kernel declined (the lean localize instantiation, or the fused score + localize
 * kernel), on the general one: a small grid strides over the list */
__global__ __launch_bounds__(64, LOC_WAVES) void pya_localize_redo_kernel(BatchDev b, const uint32_t *count,
                                                                        const uint32_t *ids, uint32_t push_cap,
                                                                        uint32_t pos_cap, uint32_t pool_cap,
                                                                        uint32_t sb, uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t n = *count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        localize_body<false>(b, ids[k], lds_raw, push_cap, pos_cap, pool_cap, sb, gtp);
        wave_lds_sync();
    }
}

/* Second pass of the lean instantiation, for launches whose first pass ran without room for the sort
 * emulation: the PSMs that pass set aside (a tie for the best PepScore), now with that room; what this
 * pass declines as well goes to the general instantiation through the second list. */
/* (its sort room holds it to two wavefronts per SIMD anyway: registers to match, no spills) */
__global__ __launch_bounds__(64, 2) void pya_localize_ties_kernel(BatchDev b, uint32_t push_cap,
                                                                              uint32_t pos_cap, uint32_t pool_cap,
                                                                              uint32_t sb, uint32_t gtp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const uint32_t n = *b.redo3_count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t psm = b.redo3_ids[k];
        const bool declined = localize_body<true>(b, psm, lds_raw, push_cap, pos_cap, pool_cap, sb, gtp, true);
        if (declined && lane_id() == 0) b.redo3b_ids[atomicAdd(b.redo3b_count, 1u)] = psm;
        wave_lds_sync();
    }
}

/* The lean localize body for PSMs score_big scored in its summary mode (thousands of site assignments): no count
 * records exist -- the signatures the body looks at are counted again (loc_recount) -- and a tie for the best
 * PepScore has been resolved by score_big (the score summary names the winner), so there is no sort.  The
 * retained-peak table and the grid score_big left are staged in LDS: recount and site-determining-ion lookups
 * stay on chip.  What the body declines (or score_big could not resolve) goes to the hand-over list. */
/* (its LDS allows four wavefronts per SIMD: 128 registers instead of the lean instantiation's 102 cost nothing) */
__global__ __launch_bounds__(64, 4) void pya_localize_recount_kernel(
    BatchDev b, const uint32_t *psm_ids, uint32_t n_ids, uint32_t cap, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap,
    uint32_t sb, uint32_t gtp, uint32_t *redo_count, uint32_t *redo_ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (blockIdx.x >= n_ids) return;
    const uint32_t psm = psm_ids[xcd_slot(blockIdx.x, n_ids)];
    const int lane = lane_id();
    /* cap = 0: the table and the grid stay in the workspace (lookups are a small share of this kernel, and its time
     * follows its occupancy: 13.3 KB = 12 wavefronts per CU with the table staged, 9.9 KB = 16 without) */
    PeakEntry *t_e = (PeakEntry *)lds_raw;
    uint16_t *grid = (uint16_t *)(lds_raw + (cap ? ((size_t)cap + PYA_TABLE_PAD) * 8 : 0));
    uint32_t *rec_batch = (uint32_t *)(grid + (cap ? PYA_GRID_CELLS : 0));
    uint32_t *hist = rec_batch + PYA_LOC_SB_MAX * PYA_REC_WORDS;
    unsigned char *rest = (unsigned char *)(hist + PYA_LOC_SB_MAX * PYA_NTOP);
    bool declined = true;
    const bool ok = b.status[psm] == PYA_ST_OK;
    const uint32_t *top = b.ws_top + (size_t)psm * 4;
    InlineSrc in;
    in.ws = b.ws + b.sig_off[psm];
    in.kmax = top[0];
    in.best_i = top[2];
    in.rec_batch = rec_batch;
    in.hist = hist;
    in.valid = ok;
    in.cand = top[3] == 0xC0DE0001u;                       /* (score_big.hip: BIG_CAND_FLAG) */
    in.tab = PeakTable();
    if (!ok || top[1] == 1u) {
        if (ok && !cap) {
            PeakTable tab;
            global_peak_table(b, psm, &tab);
            in.tab = tab;
        } else if (ok) {
            PeakTable tab;
            stage_peak_table(b, psm, t_e, &tab);
            ((uint64_t *)grid)[lane] = ((const uint64_t *)(b.grid + (size_t)psm * PYA_GRID_CELLS))[lane];
            tab.cell = grid;
            tab.base = 0.f;
            tab.inv_w = 0.f;
            tab.nb = 0.f;
            tab.last_cell = 0;
            wave_lds_sync();
            if (tab.n > 0) grid_params(&tab, t_e[0].mz, t_e[tab.n - 1].mz);
            in.tab = tab;
        }
        declined = localize_body<true>(b, psm, rest, push_cap, pos_cap, pool_cap, sb, gtp, true, in);
    }
    if (declined && lane == 0) redo_ids[atomicAdd(redo_count, 1u)] = psm;
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
    const SortLds srt = sort_carve(lds_raw, N);
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
                                   uint32_t sb, uint32_t gtp, uint32_t plain, uint32_t sort_room, hipStream_t stream) {
    if (n_ids == 0) return 0;
    const size_t lds = pya_localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
    hipError_t e;
    if (!plain) {
        e = PYA_ENSURE_MAX_LDS(pya_localize_kernel<false>);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(pya_localize_kernel<false>, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, push_cap,
                           pos_cap, pool_cap, sb, gtp, 1u);
        return (int)hipGetLastError();
    }
    /* lean instantiation first, then whatever it declined on the general one */
    e = hipMemsetAsync(b->redo3_count, 0, 2 * sizeof(uint32_t), stream);      /* both lists' counts */
    if (e != hipSuccess) return (int)e;
    e = PYA_ENSURE_MAX_LDS(pya_localize_kernel<true>);
    if (e != hipSuccess) return (int)e;
    /* sort_room = 0 (the host's choice for thousands of signatures): the lean launch without room for the sort
     * emulation; PSMs with a tie at the top go through the hand-over list to a second lean pass that has it */
    const size_t lds_lean = localize_lean_lds_bytes(push_cap, sort_room ? n_cap : 0u, pos_cap, pool_cap, sb, b->max_k);   /* (r06: its own, smaller layout) */
    hipLaunchKernelGGL(pya_localize_kernel<true>, dim3(n_ids), dim3(64), lds_lean, stream, *b, d_ids, n_ids, push_cap,
                       pos_cap, pool_cap, sb, gtp, sort_room);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = n_ids < 8192u ? n_ids : 8192u;
    const uint32_t *count = b->redo3_count, *ids = b->redo3_ids;
    if (!sort_room) {
        e = PYA_ENSURE_MAX_LDS(pya_localize_ties_kernel);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(pya_localize_ties_kernel, dim3(grid), dim3(64), lds, stream, *b, push_cap, pos_cap, pool_cap, sb,
                           gtp);
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        count = b->redo3b_count;
        ids = b->redo3b_ids;
    }
    e = PYA_ENSURE_MAX_LDS(pya_localize_redo_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_localize_redo_kernel, dim3(grid), dim3(64), lds, stream, *b, count, ids, push_cap, pos_cap,
                       pool_cap, sb, gtp);
    return (int)hipGetLastError();
}

/* test-only: the wavefront primitives of device_common.hip.h on 64 values (include/pyascore_debug.h: pya_debug_wave_ops) */
__global__ __launch_bounds__(64) void pya_debug_wave_ops_kernel(const int32_t *in, int32_t *out) {
    const int lane = lane_id();
    const int v = in[lane];
    int total;
    out[lane] = wave_excl_scan_i32(v, &total);
    out[64 + lane] = (int32_t)wave_incl_scan_u32<true>((uint32_t)v);
    out[128 + lane] = (int32_t)wave_incl_scan_u32<false>((uint32_t)v);
    out[192 + lane] = mask_rank(__ballot(v & 1));
    const int s = wave_sum_i32(v);
    const uint32_t mx = wave_max_u32((uint32_t)v), mn = wave_min_u32((uint32_t)v);
    const float fx = wave_max_f32(__int_as_float(v)), fn = wave_min_f32(__int_as_float(v));
    const uint32_t first_odd = wave_min_u32((v & 1) ? (uint32_t)lane : 0xffffffffu);
    if (lane == 17) {                                        /* (the values are wave-uniform: any lane) */
        out[256] = total;
        out[257] = s;
        out[258] = (int32_t)mx;
        out[259] = (int32_t)mn;
        out[260] = __float_as_int(fx);
        out[261] = __float_as_int(fn);
        out[262] = (int32_t)first_odd;
    }
}
extern "C" int pya_launch_debug_wave_ops(const int32_t *d_in, int32_t *d_out, hipStream_t stream) {
    hipLaunchKernelGGL(pya_debug_wave_ops_kernel, dim3(1), dim3(64), 0, stream, d_in, d_out);
    return (int)hipGetLastError();
}

/* The gather record of a PSM (pyascore_amd/shard.py: record_width = 4 + 3 k int32 words): best_score bits, n_sig, best_sig
 * lo / hi, then k Ascore bit patterns and k alternative-site masks lo / hi -- packed straight into the caller's send
 * buffer, a word per thread, rows read and written coalesced (the widest row is 4 + 3 * 64 words). */
__global__ __launch_bounds__(256) void pya_pack_records_kernel(const float *best_score, const int32_t *n_sig, const uint32_t *best_sig,
                                                               const uint32_t *ascores, const uint32_t *alt_mask, uint32_t k,
                                                               uint32_t res_k, uint64_t n_psm, uint32_t *out) {
    const uint32_t width = 4u + 3u * k;
    const uint64_t total = n_psm * width;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t psm = i / width;
        const uint32_t w = (uint32_t)(i - psm * width);
        uint32_t v;
        if (w == 0) v = __float_as_uint(best_score[psm]);
        else if (w == 1) v = (uint32_t)n_sig[psm];
        else if (w < 4) v = best_sig[2 * psm + (w - 2)];
        else if (w < 4 + k) v = (w - 4) < res_k ? ascores[psm * res_k + (w - 4)] : 0u;
        else {
            const uint32_t j = w - 4 - k;                    /* word j of the row of k 64-bit masks */
            v = (j >> 1) < res_k ? alt_mask[2 * (psm * res_k) + j] : 0u;
        }
        out[i] = v;
    }
}
extern "C" int pya_launch_pack_records(const float *best_score, const int32_t *n_sig, const uint64_t *best_sig, const float *ascores,
                                       const uint64_t *alt_mask, uint32_t k, uint32_t res_k, uint64_t n_psm, int32_t *out,
                                       hipStream_t stream) {
    if (n_psm == 0) return 0;
    const uint64_t total = n_psm * (4u + 3u * k);
    const uint32_t blocks = (uint32_t)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pya_pack_records_kernel, dim3(blocks), dim3(256), 0, stream, best_score, n_sig, (const uint32_t *)best_sig,
                       (const uint32_t *)ascores, (const uint32_t *)alt_mask, k, res_k, n_psm, (uint32_t *)out);
    return (int)hipGetLastError();
}

extern "C" size_t pya_localize_hash_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc,
                                              uint32_t hs, uint32_t pp, uint32_t max_k, uint32_t n_nl) {
    return localize_hash_lds_bytes(push_cap, n_cap, pos_cap, sb, vc, hs, pp, max_k, n_nl);
}

/* general PSMs: the hash route first, then whatever it declined on the list-based instantiation */
extern "C" int pya_launch_localize_hash(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t push_cap,
                                        uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp,
                                        uint32_t vc, uint32_t hs, uint32_t pp, uint32_t n_nl, hipStream_t stream) {
    if (n_ids == 0) return 0;
    hipError_t e = hipMemsetAsync(b->redo3_count, 0, 2 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return (int)e;
    e = PYA_ENSURE_MAX_LDS(pya_localize_hash_kernel);
    if (e != hipSuccess) return (int)e;
    const size_t lds_hash = localize_hash_lds_bytes(push_cap, n_cap, pos_cap, sb, vc, hs, pp, b->max_k, n_nl);
    hipLaunchKernelGGL(pya_localize_hash_kernel, dim3(n_ids), dim3(64), lds_hash, stream, *b, d_ids, n_ids, push_cap, pos_cap, sb,
                       vc, hs, pp);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = PYA_ENSURE_MAX_LDS(pya_localize_redo_kernel);
    if (e != hipSuccess) return (int)e;
    const size_t lds = pya_localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
    const uint32_t grid = n_ids < 8192u ? n_ids : 8192u;
    hipLaunchKernelGGL(pya_localize_redo_kernel, dim3(grid), dim3(64), lds, stream, *b, b->redo3_count, b->redo3_ids, push_cap,
                       pos_cap, pool_cap, sb, gtp);
    return (int)hipGetLastError();
}

extern "C" size_t pya_localize_recount_lds_bytes(uint32_t cap, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb) {
    return (cap ? ((size_t)cap + PYA_TABLE_PAD) * 8 + PYA_GRID_CELLS * 2 : 0) + PYA_LOC_SB_MAX * (PYA_REC_WORDS + PYA_NTOP) * 4 +
           pya_localize_lds_bytes(push_cap, 0, pos_cap, pool_cap, sb) + 64;
}

extern "C" int pya_launch_localize_recount(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap,
                                           uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp,
                                           uint32_t *d_redo_count, uint32_t *d_redo_ids, hipStream_t stream) {
    if (n_ids == 0) return 0;
    /* (the lean layout; pya_localize_recount_lds_bytes -- what the host's feasibility test asks -- is an upper bound of it) */
    const size_t lds = (cap ? ((size_t)cap + PYA_TABLE_PAD) * 8 + PYA_GRID_CELLS * 2 : 0) + PYA_LOC_SB_MAX * (PYA_REC_WORDS + PYA_NTOP) * 4 +
                       localize_lean_lds_bytes(push_cap, 0, pos_cap, pool_cap, sb, b->max_k) + 64;
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_localize_recount_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_localize_recount_kernel, dim3(n_ids), dim3(64), lds, stream, *b, d_ids, n_ids, cap, push_cap, pos_cap,
                       pool_cap, sb, gtp, d_redo_count, d_redo_ids);
    return (int)hipGetLastError();
}

/* the hand-over list of the fused kernel (score_localize.hip) on the general instantiation */
extern "C" int pya_launch_localize_redo(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids,
                                        uint32_t n_max, uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap,
                                        uint32_t pool_cap, uint32_t sb, uint32_t gtp, hipStream_t stream) {
    if (n_max == 0) return 0;
    const size_t lds = pya_localize_lds_bytes(push_cap, n_cap, pos_cap, pool_cap, sb);
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_localize_redo_kernel);
    if (e != hipSuccess) return (int)e;
    const uint32_t grid = n_max < 8192u ? n_max : 8192u;
    hipLaunchKernelGGL(pya_localize_redo_kernel, dim3(grid), dim3(64), lds, stream, *b, d_count, d_ids, push_cap,
                       pos_cap, pool_cap, sb, gtp);
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
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_ambiguity_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_ambiguity_kernel, dim3(1), dim3(64), lds, stream, *b, psm, peak_cap, list_cap,
                       ref_bits, oth_bits, d_scores, ref_ws, oth_ws, d_out);
    return (int)hipGetLastError();
}

extern "C" int pya_launch_debug_sort(const float *d_keys, uint32_t n, uint32_t *d_perm, hipStream_t stream) {
    size_t lds = sort_lds_bytes(n) + 64;
    hipError_t e = PYA_ENSURE_MAX_LDS(pya_debug_sort_kernel);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(pya_debug_sort_kernel, dim3(1), dim3(64), lds, stream, d_keys, n, d_perm);
    return (int)hipGetLastError();
}
