/* host_one.cpp -- pya_score_one: one PSM per call through a pinned, device-mapped block (PyAscore.score). */
#include "host_internal.h"
#include <climits>

/* ------------------------------------------------------------------------------------------------------ */
/* pya_score_one: one PSM per call, lowest latency (tiny_batch.hip: pya_one_kernel)                        */
/* ------------------------------------------------------------------------------------------------------ */
namespace {
/* layout of the pinned block: [flag 64 B | status 64 B | best_score, n_sig, best_sig 64 B | ascores 64 x 4 |
 * alt masks 64 x 8 | m/z PYA_FAST_PEAKS x 8 | intensity PYA_FAST_PEAKS x 8] */
const size_t kOneFlag = 0, kOneStatus = 64, kOneBest = 128, kOneAsc = 192, kOneAlt = 448, kOneMz = 1024,
             kOneInt = kOneMz + (size_t)PYA_FAST_PEAKS * 8, kOneBytes = kOneInt + (size_t)PYA_FAST_PEAKS * 8;

int one_prepare(pya_handle *h, uint32_t n_sig) {
    pya_handle::One &o = h->one;
    if (!o.host) {
        HIPCHK(h, hipHostMalloc((void **)&o.host, kOneBytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(h, hipHostGetDevicePointer((void **)&o.host_dev, o.host, 0));
        std::memset(o.host, 0, 1024);
        HIPCHK(h, hipStreamCreateWithFlags(&o.stream, hipStreamNonBlocking));
    }
    if (!o.ws.p || o.sig_cap < n_sig) {
        /* device workspace of one PSM: small arrays, the retained table, grid, per-signature scores / records / order */
        const uint32_t cap = std::max<uint32_t>(1024, next_pow2(n_sig));
        const size_t bytes = 4096 + ((size_t)PYA_FAST_PEAKS + 8) * sizeof(PeakEntry) + PYA_GRID_CELLS * 2 + 4096 +
                             (size_t)cap * (4 + PYA_REC_WORDS * 4 + 4) + 1024;
        HIPCHK(h, hipStreamSynchronize(o.stream));
        /* the retained view (if any) points into the allocation that goes away */
        if (o.view && h->kept == o.view) h->kept = nullptr;
        o.last_keep = false;
        HIPCHK(h, o.ws.alloc(bytes));
        if (h->kn.slow_null_stream) {
            /* (test switch PYA_SLOW_NULL_STREAM: 256 MB of null-stream memset in front, so that anything queued on the null
             * stream after it starts ~100 us late -- the window of the race described below made wide enough that a
             * regression is seen at once: tests/test_gpu_handover_stress.py) */
            if (!o.probe.p) HIPCHK(h, o.probe.alloc((size_t)256 << 20));
            HIPCHK(h, hipMemset(o.probe.p, 0, (size_t)256 << 20));
        }
        /* ON THE KERNEL'S OWN STREAM, and waited for: hipMemset is asynchronous to the host for device memory, it runs on
         * the null stream, and o.stream is a non-blocking stream -- a null-stream memset could still be zeroing the workspace
         * while the first kernel launched after it was already reading the scalars its lane 0 had put there (N and the
         * site count read as 0: an "unambiguous" PSM with no site assignments, status OK -- the empty result seen once in
         * r04 and once in r05; scripts/handover_repro.py, profiles/r05_handover_race.md) */
        HIPCHK(h, hipMemsetAsync(o.ws.p, 0, bytes, o.stream));
        HIPCHK(h, hipStreamSynchronize(o.stream));
        o.sig_cap = cap;
        unsigned char *w = o.ws.p;
        BatchDev &d = o.dev;
        std::memset(&d, 0, sizeof d);
        size_t at = 0;
        auto take = [&](size_t n) { unsigned char *q = w + at; at += (n + 255) & ~(size_t)255; return q; };
        d.peak_off = (const int64_t *)take(16);
        d.pep_off = (const int64_t *)take(16);
        d.aux_off = (const int64_t *)take(16);
        d.sig_off = (const int64_t *)take(16);
        d.ret_off = (const int64_t *)take(16);
        d.pep = (const uint8_t *)take(PYA_MAX_L);
        d.n_of_mod = (const int32_t *)take(4);
        d.max_charge = (const int32_t *)take(4);
        d.aux_pos = (const uint32_t *)take(PYA_ONE_MAX_AUX * 4);
        d.aux_mass = (const float *)take(PYA_ONE_MAX_AUX * 4);
        d.n_sites = (const uint8_t *)take(4);
        d.n_sig = (const uint32_t *)take(4);
        d.order_off = (const uint32_t *)take(4);
        d.desc = (const uint64_t *)take(PYA_DESC_WORDS * 8);
        d.status = (int32_t *)take(4);
        d.ret_n = (uint32_t *)take(4);
        d.ws_top = (uint32_t *)take(16);
        uint32_t *redo = (uint32_t *)take(4 * 80);
        d.redo_count = redo;
        d.redo_ids = redo + 64;
        d.redo3_count = redo + 66;
        d.redo3_ids = redo + 70;
        d.redo3b_count = redo + 67;
        d.redo3b_ids = redo + 72;
        d.redo4_count = redo + 68;
        d.redo4_ids = redo + 74;
        d.grid = (uint16_t *)take(PYA_GRID_CELLS * 2);
        d.ret = (PeakEntry *)take(((size_t)PYA_FAST_PEAKS + 8) * sizeof(PeakEntry));
        d.ws = (float *)take((size_t)cap * 4);
        d.rec = (uint32_t *)take((size_t)cap * PYA_REC_WORDS * 4);
        d.sorted_idx = (uint32_t *)take((size_t)cap * 4);
        d.mz = (const double *)(o.host_dev + kOneMz);
        d.inten = (const double *)(o.host_dev + kOneInt);
        d.best_score = (float *)(o.host_dev + kOneBest);
        d.n_sig_out = (int32_t *)(o.host_dev + kOneBest + 8);
        d.best_sig = (uint64_t *)(o.host_dev + kOneBest + 16);
        d.ascores = (float *)(o.host_dev + kOneAsc);
        d.alt_mask = (uint64_t *)(o.host_dev + kOneAlt);
    }
    return PYA_OK;
}

/* launches the kernel for o.meta (the spectrum is in the pinned block) and waits for its results */
int one_run(pya_handle *h, bool keep, uint32_t max_k) {
    pya_handle::One &o = h->one;
    const OneMeta &m = o.meta;
    BatchDev d = o.dev;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
    d.max_k = max_k;
    d.keep = keep ? 1u : 0u;
    d.debug = h->kn.debug & 0xffffu;
    /* caps of this one PSM (the rules of the plan's buckets, for a bucket of one) */
    Bucket bk;
    bk.take_knobs(h->kn);
    const uint32_t L = m.L, z = (uint32_t)m.max_charge, k = (uint32_t)m.n_of_mod, ns = m.n_sites, N = m.n_sig;
    const uint32_t n_uniq = (uint32_t)h->cfg.n_uniq, n_types = (uint32_t)h->cfg.n_types;
    const uint32_t per_type = (L - 1) * z * n_uniq;
    bk.n_cap = N;
    bk.list_cap = next_pow2(std::max<uint32_t>(per_type, 1));
    bk.pos_cap = std::max<uint32_t>(L - 1, 1);
    bk.n_types = n_types;
    bk.k_max = std::max<uint32_t>(k, 1);
    bk.push_max = std::max<uint32_t>(k < ns ? k * (ns - k) : 1, 1);
    bk.z_max = z;
    const uint32_t cap = (m.n_peaks + 31u) & ~31u;
    const uint32_t prefix = (N >= 128 && !h->kn.no_prefix) ? 1u : 0u;
    const bool plain_types = h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1;
    const uint32_t compact = (plain_types && z == 1) ? 1u : 0u;
    const bool both = h->cfg.n_fwd > 0 && h->cfg.n_fwd < h->cfg.n_types;
    uint32_t use_fused = 0, f_n_cap = 4, f_stride = 8, f_ent = 1, f_push = 8;
    const uint32_t frags = (both ? 2u : 1u) * (L - 1) * z;
    if (!keep && plain_types && !h->kn.no_fused && k < ns && N > 0 && N <= (both ? 32u : 64u) && frags <= 255u) {
        use_fused = both ? 1u : 2u;
        f_n_cap = (N + 3u) & ~3u;
        f_stride = (both ? 2u : 1u) * f_n_cap + 4u;
        f_ent = std::max<uint32_t>((L - 1) * z, 1);
        f_push = std::max<uint32_t>(8u, bk.push_cap());
    }
    if (pya_one_lds_bytes(cap, prefix, h->cfg.n_nl != 0, compact, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb(), use_fused,
                          f_n_cap, f_stride, f_ent, f_push, z > 1) > kMaxLds)
        return PYA_ERR_STATE;        /* no room in one workgroup's LDS: declined, the caller takes the batch path (per-stage kernels) */
    volatile uint32_t *flag = (volatile uint32_t *)(o.host + kOneFlag);
    /* belt and braces for the hand-over through host memory: the kernel echoes the sequence number next to the status
     * before it publishes it, and n_sig starts from a value no result has -- the results are read only when all three
     * say this call's kernel wrote them */
    volatile int32_t *echo = (volatile int32_t *)(o.host + kOneStatus + 4);
    volatile int32_t *n_sig_word = (volatile int32_t *)(o.host + kOneBest + 8);
    const int32_t kUnset = INT32_MIN;
    *n_sig_word = kUnset;
    const auto t_launch = std::chrono::steady_clock::now();
    int e = pya_launch_one(&d, &m, cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(),
                           bk.sb(), bk.gtp(), use_fused, f_n_cap, f_stride, f_ent, f_push, z > 1 ? 1u : 0u,
                           (int32_t *)(o.host_dev + kOneStatus), (uint32_t *)(o.host_dev + kOneFlag), o.stream);
    if (e) return h->hip_fail((hipError_t)e, "score (one PSM) launch");
    /* the kernel's last store is the sequence number: poll it (a stream synchronisation costs several
     * microseconds more); give up after two seconds and ask the runtime what happened */
    const auto t0 = std::chrono::steady_clock::now();
    o.t_sum[2] += std::chrono::duration<double>(t0 - t_launch).count();
    auto published = [&]() { return *flag == m.seq && (uint32_t)*echo == m.seq && *n_sig_word != kUnset; };
    for (uint64_t spins = 0; !published(); spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HIPCHK(h, hipStreamSynchronize(o.stream));
            if (!published()) return h->fail(PYA_ERR_HIP, 0, "the kernel finished without publishing its results");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    o.t_sum[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    o.last_keep = keep;
    o.last_max_k = max_k;
    /* a retained view for pya_get_pep_scores / pya_calculate_ambiguity */
    if (keep) {
        if (!o.view) o.view = new pya_plan;
        pya_plan *p = o.view;
        p->h = h;
        p->flags = PYA_FLAG_KEEP;
        p->n_psm = 1;
        p->peak_cap = cap;
        p->max_k = max_k;
        p->sig_off = {0, (int64_t)N};
        p->pep_off = {0, (int64_t)L};
        p->n_sig = {N};
        p->order_off = {m.order_off};
        p->max_charge = {m.max_charge};
        p->d_rec.adopt(d.rec, (size_t)N * PYA_REC_WORDS);
        p->d_sorted.adopt(d.sorted_idx, N);
        p->d_ws.adopt(d.ws, N);
        p->dev = d;
        p->quiesced = true;
        p->ran = true;
        p->last_stream = o.stream;
        if (h->kept && h->kept != p) pya_plan_destroy(h->kept);
        h->kept = p;
    } else if (h->kept && h->kept == o.view) {
        h->kept = nullptr;                                  /* the view described the previous PSM */
    }
    return PYA_OK;
}
}  // namespace

extern "C" int pya_score_one(pya_handle *h, const double *mz, const double *inten, uint64_t n_peaks, const uint8_t *pep,
                             uint64_t L, int32_t n_of_mod, int32_t max_charge, const uint32_t *aux_pos, const float *aux_mass,
                             uint64_t n_aux, uint32_t flags, const pya_results *out) {
    if (!h || !out) return PYA_ERR_ARG;
    const auto t_in = std::chrono::steady_clock::now();
    h->err.clear();
    h->err_index = -1;
    h->last_status.clear();
    if (h->kn.no_tiny) return PYA_ERR_STATE;               /* (route switch of the tests: the kernel-per-stage path) */
    if (!mz || !inten || !pep) return h->fail(PYA_ERR_ARG, -1, "NULL array");
    if (!out->best_score || !out->best_sig || !out->n_sig || !out->ascores || !out->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in results");
    if (n_aux && (!aux_pos || !aux_mass)) return h->fail(PYA_ERR_ARG, -1, "NULL fixed-modification arrays");
    /* validation: what plan_create_impl checks for a PSM (same messages) */
    if (n_peaks == 0) return h->fail(PYA_ERR_PSM, 0, "PSM 0: empty spectrum");
    if (n_peaks > PYA_MAX_PEAKS) return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %llu peaks exceed the limit of %d", (unsigned long long)n_peaks, PYA_MAX_PEAKS);
    if (L < 1 || L > PYA_MAX_PEPTIDE_LEN)
        return h->fail(L < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, 0, "PSM 0: peptide length %lld outside 1..%d", (long long)L, PYA_MAX_PEPTIDE_LEN);
    if (n_of_mod < 0) return h->fail(PYA_ERR_PSM, 0, "PSM 0: negative n_of_mod");
    if (max_charge < 1 || max_charge > PYA_MAX_CHARGE)
        return h->fail(max_charge < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, 0, "PSM 0: max_fragment_charge %d outside 1..%d", max_charge, PYA_MAX_CHARGE);
    uint32_t ns = 0;
    for (uint64_t j = 0; j < L; j++) {
        if (!h->is_residue[pep[j]])
            return h->fail(PYA_ERR_PSM, 0, "PSM 0: unknown residue '%c' at position %lld", (char)pep[j], (long long)(j + 1));
        if (h->letter_modifiable((char)pep[j], (size_t)j, (size_t)L)) ns++;
    }
    for (uint64_t a = 0; a < n_aux; a++)
        if (aux_pos[a] > (uint32_t)L) return h->fail(PYA_ERR_PSM, 0, "PSM 0: aux_mod_pos %u beyond the peptide", aux_pos[a]);
    if (ns > PYA_MAX_SITES) return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %u modifiable residues exceed %d", ns, PYA_MAX_SITES);
    uint64_t N = 0;
    if ((uint32_t)n_of_mod <= ns) {
        uint64_t &cached = h->binom_cache[ns][n_of_mod];
        if (cached == 0) cached = binom(ns, (uint32_t)n_of_mod);
        N = cached;
    }
    if (N > PYA_MAX_SIGNATURES)
        return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: C(%u,%d) site assignments exceed the limit of %d", ns, n_of_mod, PYA_MAX_SIGNATURES);
    const uint32_t per_type = (uint32_t)(L - 1) * (uint32_t)max_charge * (uint32_t)h->cfg.n_uniq;
    if (per_type > PYA_MAX_FRAGMENTS_PER_TYPE)
        return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %u fragments per ion type exceed %d", per_type, PYA_MAX_FRAGMENTS_PER_TYPE);
    /* (beyond a limit of the fast kernels: the caller takes the batch path, which has the general kernel) */
    if (h->all_general() || n_peaks > PYA_FAST_PEAKS || L > PYA_FAST_PEPTIDE_LEN || N > PYA_FAST_SIGNATURES || per_type > PYA_FAST_FRAGMENTS_PER_TYPE) return PYA_ERR_STATE;
    if (n_aux > PYA_ONE_MAX_AUX || (uint32_t)n_of_mod > 64u) return PYA_ERR_STATE;     /* (the caller takes the batch path) */
    if (out->max_k < (uint32_t)std::max(n_of_mod, 1)) return h->fail(PYA_ERR_ARG, -1, "results.max_k is smaller than n_of_mod");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_config(h);
    if (rc) return rc;
    uint32_t ooff = 0;
    if (N) {
        uint32_t &co = h->shape_cache[ns][n_of_mod];
        if (co == 0xffffffffu) co = shape_offset(h, ns, (uint32_t)n_of_mod);
        ooff = co;
    }
    rc = ensure_lut(h, per_type * (uint32_t)h->cfg.n_types);
    if (rc) return rc;
    if (h->order_uploaded != h->order_tab.size() || !h->d_order.p) {
        HIPCHK(h, h->d_order.upload(h->order_tab.data(), h->order_tab.size()));
        HIPCHK(h, h->d_inv.upload(h->inv_tab.data(), h->inv_tab.size()));
        if (!h->d_binom.p) {
            std::vector<uint32_t> bt(64 * 64);
            for (uint32_t pp = 0; pp < 64; pp++)
                for (uint32_t t = 0; t < 64; t++) bt[pp * 64 + t] = (uint32_t)std::min<uint64_t>(binom(pp, t), 0xffffffffull);
            HIPCHK(h, h->d_binom.upload(bt.data(), bt.size()));
        }
        HIPCHK(h, hipDeviceSynchronize());
        h->order_uploaded = h->order_tab.size();
    }
    rc = one_prepare(h, (uint32_t)N);
    if (rc) return rc;
    pya_handle::One &o = h->one;
    /* inputs: the spectrum into the pinned block, everything else into the kernel's arguments */
    const auto t_copy = std::chrono::steady_clock::now();
    std::memcpy(o.host + kOneMz, mz, (size_t)n_peaks * 8);
    std::memcpy(o.host + kOneInt, inten, (size_t)n_peaks * 8);
    OneMeta &m = o.meta;
    std::memset(&m, 0, sizeof m);
    std::memcpy(m.pep, pep, (size_t)L);
    m.n_peaks = (uint32_t)n_peaks;
    m.L = (uint32_t)L;
    m.n_aux = (uint32_t)n_aux;
    m.n_sig = (uint32_t)N;
    m.order_off = ooff;
    m.seq = ++o.seq ? o.seq : ++o.seq;
    m.n_of_mod = n_of_mod;
    m.max_charge = max_charge;
    m.n_sites = ns;
    for (uint64_t a = 0; a < n_aux; a++) {
        m.aux_pos[a] = aux_pos[a];
        m.aux_mass[a] = aux_mass[a];
    }
    m.desc[0] = 0;
    m.desc[1] = 0;
    m.desc[2] = 0;
    m.desc[3] = 0;
    m.desc[4] = (uint64_t)L | (uint64_t)n_aux << 16 | (uint64_t)((uint32_t)n_of_mod & 0xffffu) << 32 | (uint64_t)ns << 48 |
                (uint64_t)((uint32_t)max_charge & 0xffu) << 56;
    m.desc[5] = (uint64_t)N | (uint64_t)ooff << 32;
    o.have_last = true;
    const uint32_t mk = out->max_k;
    if (mk > 64) return PYA_ERR_STATE;
    const auto t_run = std::chrono::steady_clock::now();
    o.t_sum[0] += std::chrono::duration<double>(t_copy - t_in).count();
    o.t_sum[1] += std::chrono::duration<double>(t_run - t_copy).count();
    rc = one_run(h, (flags & PYA_FLAG_KEEP) != 0, mk);
    if (rc) return rc;
    const auto t_out = std::chrono::steady_clock::now();
    const int32_t st = *(const int32_t *)(o.host + kOneStatus);
    rc = check_status(h, &st, 1, false);
    if (rc) return rc;
    out->best_score[0] = *(const float *)(o.host + kOneBest);
    out->n_sig[0] = *(const int32_t *)(o.host + kOneBest + 8);
    out->best_sig[0] = *(const uint64_t *)(o.host + kOneBest + 16);
    std::memcpy(out->ascores, o.host + kOneAsc, (size_t)mk * 4);
    std::memcpy(out->alt_mask, o.host + kOneAlt, (size_t)mk * 8);
    o.t_sum[4] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_out).count();
    for (int i = 0; i < 4; i++) o.t_dev[i] += 1e-8 * (double)*(const int32_t *)(o.host + kOneStatus + 8 + 4 * i);   /* 100 MHz ticks */
    o.t_cycles += (double)*(const int32_t *)(o.host + kOneStatus + 24);
    o.t_calls++;
    return PYA_OK;
}

/* diagnostics: average microseconds per pya_score_one call since the last call of this function, by stage --
 * checks and tables, copying the spectrum into the pinned block, the launch call (with the caps before it), waiting
 * for the kernel's flag, copying the results out; us[5] = calls averaged; us[6..9] = inside the kernel (its own
 * 100 MHz clock): scalars into place, binning, scoring (+ localisation on the fused route), the rest; us[10] = shader
 * clock cycles of the kernel (over the sum of 6..9: the clock it ran at) */
extern "C" int pya_one_times(pya_handle *h, double us[12]) {
    if (!h || !us) return PYA_ERR_ARG;
    pya_handle::One &o = h->one;
    for (int i = 0; i < 12; i++) us[i] = 0.;
    for (int i = 0; i < 5; i++) {
        us[i] = o.t_calls ? 1e6 * o.t_sum[i] / (double)o.t_calls : 0.;
        o.t_sum[i] = 0.;
    }
    for (int i = 0; i < 4; i++) {
        us[6 + i] = o.t_calls ? 1e6 * o.t_dev[i] / (double)o.t_calls : 0.;
        o.t_dev[i] = 0.;
    }
    us[5] = (double)o.t_calls;
    us[10] = o.t_calls ? o.t_cycles / (double)o.t_calls : 0.;       /* shader clock cycles per kernel */
    o.t_cycles = 0.;
    o.t_calls = 0;
    return PYA_OK;
}

/* the last pya_score_one PSM once more, retained (PYA_FLAG_KEEP): what PyAscore.pep_scores and
 * calculate_ambiguity need; the spectrum and the scalars are still where the last call put them */
extern "C" int pya_rescore_last_keep(pya_handle *h) {
    if (!h) return PYA_ERR_ARG;
    if (!h->one.have_last) return h->fail(PYA_ERR_STATE, -1, "no PSM scored with pya_score_one yet");
    if (h->one.last_keep && h->kept == h->one.view) return PYA_OK;
    HIPCHK(h, hipSetDevice(h->device));
    h->one.meta.seq = ++h->one.seq ? h->one.seq : ++h->one.seq;
    return one_run(h, true, h->one.last_max_k);
}

