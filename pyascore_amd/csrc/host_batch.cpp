/* host_batch.cpp -- pya_score_batch: host arrays in, host results out; chunking and the upload / kernels / download pipeline. */
#include "host_internal.h"

namespace {

const size_t kChunkMin = 32u << 20;          /* spectra bytes below which a call is not worth pipelining */
const size_t kChunkTarget = 96u << 20;       /* spectra bytes per chunk when the budget allows more       */
const size_t kDefaultBudget = (size_t)6 << 30;

}  // namespace

size_t workspace_budget(const pya_handle *h) {
    if (h->ws_budget) return h->ws_budget;
    if (h->kn.workspace_mb > 0) return (size_t)std::max<int64_t>(16, h->kn.workspace_mb) << 20;
    return kDefaultBudget;
}

namespace {

/* Device bytes a chunk [lo, hi) holds while it is scored: its spectra in the upload ring (two
 * slots, so twice) and its arena (retained table, grid, per-signature scores and records, results
 * and metadata).  C(n,k) comes from the peptide letters, as in the plan's pre-pass. */
struct ChunkCost {
    std::vector<double> arena, io;             /* per PSM */
    std::vector<uint8_t> sites;                /* modifiable residues per PSM, 255 = invalid letters / length */
};
ChunkCost chunk_costs(pya_handle *h, const pya_batch *b, uint32_t max_k) {
    const uint64_t n = b->n_psm;
    ChunkCost c;
    c.arena.resize(n);
    c.io.resize(n);
    c.sites.assign(n, 255);
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const int64_t P = std::max<int64_t>(0, b->peak_off[i + 1] - b->peak_off[i]);
            const int64_t L = b->pep_off[i + 1] - b->pep_off[i];
            double sigs = 0;
            if (L >= 1 && L <= PYA_MAX_PEPTIDE_LEN) {
                uint32_t ns = 0;
                const bool ok = h->scan_peptide(b->pep + b->pep_off[i], L, &ns);
                if (ok && ns < 255u) c.sites[i] = (uint8_t)ns;
                uint64_t N = 0;
                if (ok && ns <= PYA_MAX_SITES && b->n_of_mod[i] >= 0 && (uint32_t)b->n_of_mod[i] <= ns) {
                    uint64_t &cached = h->binom_cache[ns][b->n_of_mod[i]];      /* benign race: same value */
                    if (cached == 0) cached = binom(ns, (uint32_t)b->n_of_mod[i]);
                    N = cached;
                }
                sigs = N > PYA_MAX_SIGNATURES ? 0. : (double)N;
            }
            c.io[i] = 16.0 * (double)P;
            /* per site assignment: PepScore 4 + count record 4 x rec_words (6 for n_top = 10, 9 for 16); a PSM beyond the fast
             * kernels' limits (or any PSM of a scorer with n_top > 10) also has its slice of the general kernel's scratch, a
             * spectrum of more than PYA_FAST_PEAKS peaks the global binning kernel's 15 bytes per peak (sized by the largest) */
            double extra = 0.;
            const uint32_t per_type = (uint32_t)std::max<int64_t>(L - 1, 0) * (uint32_t)std::max(1, std::min(b->max_charge[i], PYA_MAX_CHARGE)) * (uint32_t)h->cfg.n_uniq;
            if (P > PYA_FAST_PEAKS || L > PYA_FAST_PEPTIDE_LEN || sigs > PYA_FAST_SIGNATURES || per_type > PYA_FAST_FRAGMENTS_PER_TYPE || h->all_general()) {
                const uint32_t ns = c.sites[i] == 255 ? 0u : c.sites[i], kk = (uint32_t)std::max(0, b->n_of_mod[i]);
                extra += (double)pya_general_scratch_bytes((uint32_t)sigs, kk <= ns ? ((kk * (ns - kk)) + 3u) & ~3u : 0u) + 256.0;
                if (P > PYA_FAST_PEAKS) extra += 16.0 * (double)PYA_MAX_PEAKS;
            }
            c.arena[i] = 8.0 * (double)(P + 1) + 8.0 + (4.0 + 4.0 * (double)h->rec_words()) * sigs + extra + (double)L + 2.0 * PYA_GRID_CELLS + 96.0 + 12.0 * max_k;
        }
    };
    unsigned nt = n >= 20000 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
    return c;
}

/* "PSM 12: ..." of a chunk that starts at PSM `lo` of the caller's batch -> "PSM <12 + lo>: ..." */
void rebase_error(pya_handle *h, uint64_t lo) {
    if (h->err_index >= 0) h->err_index += (int64_t)lo;
    unsigned long long local = 0;
    int used = 0;
    if (lo && std::sscanf(h->err.c_str(), "PSM %llu%n", &local, &used) == 1)
        h->err = "PSM " + std::to_string(local + lo) + h->err.substr((size_t)used);
}

}  // namespace

/* Big pya_score_batch calls: the batch is cut into chunks of consecutive PSMs that fit the device
 * budget and the chunks are pipelined -- a helper thread streams the spectra of chunk c + 1 over
 * PCIe (the bound of this entry point: 16 bytes per peak) into the other slot of a two-slot ring
 * while this thread plans chunk c, runs its kernels and brings its results back on a second
 * stream.  A call of any size completes; it never fails for lack of workspace. */
static int score_batch_chunked(pya_handle *h, const pya_batch *b, const double *mz, const double *inten,
                               uint32_t flags, const pya_results *out, const std::vector<uint64_t> &cuts,
                               const uint8_t *pre_sites) {
    const size_t nchunk = cuts.size() - 1;
    const uint32_t mk = out->max_k;
    const bool skip = (flags & PYA_FLAG_SKIP_INVALID) != 0;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->run_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->run_stream, hipStreamNonBlocking));
    size_t slot_peaks = 0;
    for (size_t c = 0; c < nchunk; c++)
        slot_peaks = std::max<size_t>(slot_peaks, (size_t)(b->peak_off[cuts[c + 1]] - b->peak_off[cuts[c]]));
    for (auto &slot : h->io_ring)
        if (slot.n < slot_peaks * 2) HIPCHK(h, slot.alloc(slot_peaks * 2));
    if (skip) h->last_status.assign(b->n_psm, 0);

    /* uploader: chunk c may be written once chunk c - 2 has been consumed */
    std::mutex mu;
    std::condition_variable cv;
    size_t uploaded = 0, consumed = 0;
    bool stop = false;
    hipError_t up_err = hipSuccess;
    const int device = h->device;
    std::thread uploader([&]() {
        hipError_t e = hipSetDevice(device);
        for (size_t c = 0; c < nchunk && e == hipSuccess; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || c < consumed + 2; });
                if (stop) break;
            }
            const int64_t p0 = b->peak_off[cuts[c]], np = b->peak_off[cuts[c + 1]] - p0;
            double *dst = h->io_ring[c & 1].p;
            if (np > 0) {
                e = hipMemcpyAsync(dst, mz + p0, (size_t)np * 8, hipMemcpyHostToDevice, h->copy_stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(dst + np, inten + p0, (size_t)np * 8, hipMemcpyHostToDevice, h->copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
            }
            std::lock_guard<std::mutex> lk(mu);
            up_err = e;
            uploaded = c + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess) up_err = e;
        uploaded = nchunk;                                   /* nobody waits for chunks that will not come */
        cv.notify_all();
    });
    auto finish = [&](int rc) {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        uploader.join();
        return rc;
    };

    /* this thread: plan chunk c + 1 (host pre-pass) while the GPU runs the kernels and the result copy of
     * chunk c; two plans alive at a time */
    auto make_plan = [&](size_t c, pya_plan **pp) -> int {
        const uint64_t lo = cuts[c], hi = cuts[c + 1];
        const int64_t np = b->peak_off[hi] - b->peak_off[lo];
        pya_batch sub = *b;
        sub.n_psm = hi - lo;
        sub.peak_off = b->peak_off + lo;
        sub.pep_off = b->pep_off + lo;
        sub.n_of_mod = b->n_of_mod + lo;
        sub.max_charge = b->max_charge + lo;
        if (b->aux_off) sub.aux_off = b->aux_off + lo;
        IoReq io = {mz, inten, mk, h->io_ring[c & 1].p, h->io_ring[c & 1].p + np, h->run_stream,
                    pre_sites ? pre_sites + lo : nullptr};
        int rc = plan_create_impl(h, &sub, flags & ~(PYA_FLAG_TIMING | PYA_FLAG_KEEP), &io, pp);
        if (rc) rebase_error(h, lo);
        return rc;
    };
    typedef std::unique_ptr<pya_plan, void (*)(pya_plan *)> PlanPtr;
    pya_plan *raw = nullptr;
    int rc = make_plan(0, &raw);
    if (rc) return finish(rc);
    PlanPtr cur(raw, pya_plan_destroy), next(nullptr, pya_plan_destroy);
    for (size_t c = 0; c < nchunk; c++) {
        const uint64_t lo = cuts[c], n = cuts[c + 1] - lo;
        pya_plan *p = cur.get();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return uploaded > c; });
            if (up_err != hipSuccess) {
                lk.unlock();
                return finish(h->hip_fail(up_err, "spectrum upload"));
            }
        }
        pya_results d_out = {mk, p->d_best_score.p, p->d_best_sig.p, p->d_n_sig_out.p, p->d_ascores.p, p->d_alt.p};
        rc = pya_plan_run(p, p->d_mz.p, p->d_inten.p, h->run_stream, &d_out);
        if (rc) return finish(rc);
        /* status + results are adjacent in the arena: one asynchronous copy into pinned memory */
        void *&pin = h->pinned_stage[c & 1];
        if (h->pinned_bytes[c & 1] < p->d2h_bytes) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            h->pinned_bytes[c & 1] = 0;
            hipError_t e0 = hipHostMalloc(&pin, p->d2h_bytes + p->d2h_bytes / 4, hipHostMallocDefault);
            if (e0 != hipSuccess) return finish(h->hip_fail(e0, "pinned result buffer"));
            h->pinned_bytes[c & 1] = p->d2h_bytes + p->d2h_bytes / 4;
        }
        unsigned char *sg = (unsigned char *)pin;
        hipError_t e = hipMemcpyAsync(sg, p->arena.p + p->o_status, p->d2h_bytes, hipMemcpyDeviceToHost, h->run_stream);
        if (e != hipSuccess) return finish(h->hip_fail(e, "results copy"));
        hipEvent_t done = nullptr;                                /* chunk c finished (kernels + copy) */
        e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done, h->run_stream);
        if (e != hipSuccess) return finish(h->hip_fail(e, "event"));
        int rc_next = PYA_OK;
        if (c + 1 < nchunk) {                                     /* CPU pre-pass of the next chunk meanwhile */
            raw = nullptr;
            rc_next = make_plan(c + 1, &raw);
            next.reset(raw);
        }
        e = hipEventSynchronize(done);
        (void)hipEventDestroy(done);
        p->quiesced = e == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(mu);                   /* the chunk's ring slot may be overwritten now */
            consumed = c + 1;
            cv.notify_all();
        }
        if (e != hipSuccess) return finish(h->hip_fail(e, "results copy"));
        if (skip) std::memcpy(h->last_status.data() + lo, sg, n * sizeof(int32_t));
        const std::string keep_err = h->err;                      /* make_plan(c + 1) may have set a message */
        const int64_t keep_idx = h->err_index;
        rc = check_status(h, (const int32_t *)sg, n, skip);
        if (rc) {
            rebase_error(h, lo);
            return finish(rc);
        }
        if (rc_next) {
            h->err = keep_err;
            h->err_index = keep_idx;
            return finish(rc_next);
        }
        const size_t o = p->o_status;
        std::memcpy(out->best_score + lo, sg + (p->o_best_score - o), n * sizeof(float));
        std::memcpy(out->best_sig + lo, sg + (p->o_best_sig - o), n * sizeof(uint64_t));
        std::memcpy(out->n_sig + lo, sg + (p->o_n_sig_out - o), n * sizeof(int32_t));
        std::memcpy(out->ascores + lo * mk, sg + (p->o_ascores - o), n * mk * sizeof(float));
        std::memcpy(out->alt_mask + lo * mk, sg + (p->o_alt - o), n * mk * sizeof(uint64_t));
        cur = std::move(next);
    }
    return finish(PYA_OK);
}

int pya_score_batch(pya_handle *h, const pya_batch *b, const double *mz, const double *inten, uint32_t flags,
                    const pya_results *out) {
    if (!h || !b || !out) return PYA_ERR_ARG;
    h->last_status.clear();
    if (b->n_psm == 0) return PYA_OK;
    if (!mz || !inten) return h->fail(PYA_ERR_ARG, -1, "NULL spectrum arrays");
    if (!b->peak_off || !b->pep || !b->pep_off || !b->n_of_mod || !b->max_charge)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in batch");
    if (!out->best_score || !out->best_sig || !out->n_sig || !out->ascores || !out->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in results");
    if (b->peak_off[b->n_psm] < b->peak_off[0]) return h->fail(PYA_ERR_ARG, -1, "peak_off is not monotone");
    /* (not while the records of a pya_score_one PSM are retained in the one-PSM workspace: this call would overwrite
     * what pya_get_pep_scores / pya_calculate_ambiguity still read there) */
    const bool one_view_live = h->kept && h->kept == h->one.view;
    if (b->n_psm == 1 && !(flags & (PYA_FLAG_SKIP_INVALID | PYA_FLAG_TIMING)) && !one_view_live) {
        /* a batch of one is PyAscore.score: the low-latency path (it declines what it has no room for) */
        const bool has_aux1 = b->aux_off && b->aux_pos && b->aux_mass;
        const int64_t a0 = has_aux1 ? b->aux_off[0] : 0, a1 = has_aux1 ? b->aux_off[1] : 0;
        const int64_t P1 = b->peak_off[1] - b->peak_off[0], L1 = b->pep_off[1] - b->pep_off[0];
        if (a1 >= a0 && P1 >= 0 && L1 >= 0) {
            int rc1 = pya_score_one(h, mz + b->peak_off[0], inten + b->peak_off[0], (uint64_t)P1, b->pep + b->pep_off[0], (uint64_t)L1,
                                    b->n_of_mod[0], b->max_charge[0], has_aux1 ? b->aux_pos + a0 : nullptr,
                                    has_aux1 ? b->aux_mass + a0 : nullptr, (uint64_t)(a1 - a0), flags & PYA_FLAG_KEEP, out);
            /* the one-PSM staging now holds THIS PSM: pya_rescore_last_keep must not replay it as the caller's last
             * pya_score_one PSM (it fails with PYA_ERR_STATE instead) */
            if (!(flags & PYA_FLAG_KEEP)) h->one.have_last = false;
            if (rc1 != PYA_ERR_STATE || !h->err.empty()) return rc1;
        }
    }
    {
        /* Chunking: needed when the call does not fit the device budget, worthwhile (pipelining)
         * when there is enough PCIe traffic to hide the kernels under.  A retained batch
         * (PYA_FLAG_KEEP) stays one plan: its records are queried by PSM afterwards. */
        const size_t io_total = (size_t)(b->peak_off[b->n_psm] - b->peak_off[0]) * 16;
        if (!(flags & PYA_FLAG_KEEP) && io_total >= kChunkMin && !h->kn.no_chunks) {
            const size_t budget = workspace_budget(h);
            const ChunkCost cost = chunk_costs(h, b, out->max_k);
            double io_target = (double)kChunkTarget;
            if (h->kn.chunk_mb > 0.) io_target = h->kn.chunk_mb * 1048576.0;
            std::vector<uint64_t> cuts{0};
            double io = 0, arena = 0;
            for (uint64_t i = 0; i < b->n_psm; i++) {
                const double io2 = io + cost.io[i], ar2 = arena + cost.arena[i];
                if (i > cuts.back() && (io2 > io_target || 2.0 * io2 + ar2 > (double)budget)) {
                    cuts.push_back(i);
                    io = cost.io[i];
                    arena = cost.arena[i];
                } else {
                    io = io2;
                    arena = ar2;
                }
            }
            cuts.push_back(b->n_psm);
            if (cuts.size() > 2) return score_batch_chunked(h, b, mz, inten, flags, out, cuts, cost.sites.data());
        }
    }
    const bool host_timing = h->kn.host_timing;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pya host] %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    pya_plan *p = nullptr;
    IoReq io = {mz, inten, out->max_k, nullptr, nullptr, nullptr, nullptr};
    /* Big batches: the spectra (16 bytes per peak, PCIe-bound) go up on a helper thread while this
     * one runs the host pre-pass of the plan; small ones ride in the plan's single staged copy. */
    const int64_t peaks_lo = b->peak_off[0], n_peaks = b->peak_off[b->n_psm] - peaks_lo;
    std::thread uploader;
    hipError_t up_err = hipSuccess;
    if (n_peaks > 0 && (size_t)n_peaks * 16 > kStageLimit && !h->kn.no_upload_thread) {
        HIPCHK(h, hipSetDevice(h->device));
        if (h->io_buf.n < (size_t)n_peaks * 2) HIPCHK(h, h->io_buf.alloc((size_t)n_peaks * 2));
        io.d_mz_ext = h->io_buf.p;
        io.d_inten_ext = h->io_buf.p + n_peaks;
        const int device = h->device;
        uploader = std::thread([&, device]() {
            up_err = hipSetDevice(device);
            if (up_err == hipSuccess)
                up_err = hipMemcpy(io.d_mz_ext, mz + peaks_lo, (size_t)n_peaks * 8, hipMemcpyHostToDevice);
            if (up_err == hipSuccess)
                up_err = hipMemcpy(io.d_inten_ext, inten + peaks_lo, (size_t)n_peaks * 8, hipMemcpyHostToDevice);
        });
    }
    int rc = plan_create_impl(h, b, flags & ~PYA_FLAG_TIMING, &io, &p);
    if (uploader.joinable()) uploader.join();
    if (rc) return rc;
    if (up_err != hipSuccess) {
        pya_plan_destroy(p);
        return h->hip_fail(up_err, "spectrum upload");
    }
    std::unique_ptr<pya_plan, void (*)(pya_plan *)> guard(p, pya_plan_destroy);
    lap("plan + h2d");
    const uint64_t n = b->n_psm;
    const uint32_t mk = out->max_k;
    pya_results d_out = {mk, p->d_best_score.p, p->d_best_sig.p, p->d_n_sig_out.p, p->d_ascores.p, p->d_alt.p};
    rc = pya_plan_run(p, p->d_mz.p, p->d_inten.p, nullptr, &d_out);
    if (rc) return rc;
    if (p->d2h_bytes <= kStageLimit) {
        /* status and results are adjacent in the arena: one copy, which also waits for the kernels */
        h->stage.resize(std::max(h->stage.size(), p->d2h_bytes));
        unsigned char *sg = h->stage.data();
        HIPCHK(h, hipMemcpy(sg, p->arena.p + p->o_status, p->d2h_bytes, hipMemcpyDeviceToHost));
        lap("kernels + d2h");
        const bool skip = (flags & PYA_FLAG_SKIP_INVALID) != 0;
        if (skip) h->last_status.assign((const int32_t *)sg, (const int32_t *)sg + n);
        rc = check_status(h, (const int32_t *)sg, n, skip);
        if (rc) return rc;
        const size_t o = p->o_status;
        std::memcpy(out->best_score, sg + (p->o_best_score - o), n * sizeof(float));
        std::memcpy(out->best_sig, sg + (p->o_best_sig - o), n * sizeof(uint64_t));
        std::memcpy(out->n_sig, sg + (p->o_n_sig_out - o), n * sizeof(int32_t));
        std::memcpy(out->ascores, sg + (p->o_ascores - o), n * mk * sizeof(float));
        std::memcpy(out->alt_mask, sg + (p->o_alt - o), n * mk * sizeof(uint64_t));
    } else {
        rc = pya_plan_check(p);
        if (rc) return rc;
        lap("kernels");
        HIPCHK(h, hipMemcpy(out->best_score, p->d_best_score.p, n * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->best_sig, p->d_best_sig.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->n_sig, p->d_n_sig_out.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->ascores, p->d_ascores.p, n * mk * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->alt_mask, p->d_alt.p, n * mk * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    lap("d2h");
    if (flags & PYA_FLAG_KEEP) {
        if (h->kept) pya_plan_destroy(h->kept);
        h->kept = guard.release();
    }
    return PYA_OK;
}
