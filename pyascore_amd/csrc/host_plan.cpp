/* host_plan.cpp -- a plan: the host pre-pass over a batch (validation, routes, id lists), ONE device arena, the launches. */
#include "host_internal.h"

void refresh_shared(pya_plan *p) {
    pya_handle *h = p->h;
    BatchDev &d = p->dev;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
}

void fill_dev(pya_plan *p) {
    pya_handle *h = p->h;
    BatchDev &d = p->dev;
    std::memset(&d, 0, sizeof d);
    d.peak_off = p->d_peak_off.p;
    d.pep = p->d_pep.p;
    d.pep_off = p->d_pep_off.p;
    d.n_of_mod = p->d_n_of_mod.p;
    d.max_charge = p->d_max_charge.p;
    d.aux_pos = p->d_aux_pos.p;
    d.aux_mass = p->d_aux_mass.p;
    d.aux_off = p->d_aux_off.p;
    d.n_sites = p->d_n_sites.p;
    d.n_sig = p->d_n_sig.p;
    d.order_off = p->d_order_off.p;
    d.sig_off = p->d_sig_off.p;
    d.desc = p->d_desc.p;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
    d.ret = p->d_ret.p;
    d.ret_off = p->d_ret_off.p;
    d.ret_n = p->d_ret_n.p;
    d.grid = p->d_grid.p;
    d.redo_count = p->d_redo.p;
    d.redo_ids = p->d_redo.p + 64;
    d.redo3_count = p->d_redo3.p;
    d.redo3_ids = p->d_redo3.p + 64;
    d.redo3b_count = p->d_redo3.p + 1;
    d.redo3b_ids = p->d_redo3.p + 64 + p->n_psm;
    d.redo4_count = p->d_redo.p + 1;          /* (the hand-over counts share the head of d_redo: one memset per run) */
    d.redo4_ids = p->d_redo4.p + 64;
    d.ws = p->d_ws.p;
    d.ws_top = p->d_ws_top.p;
    d.rec = p->d_rec.p;
    d.sorted_idx = p->d_sorted.p;
    d.status = p->d_status.p;
    d.max_k = p->max_k;
    d.keep = (p->flags & PYA_FLAG_KEEP) ? 1u : 0u;
    d.debug = h->kn.debug;
    if (h->kn.stamps) {
        if (!p->d_stamps.p) {
            (void)p->d_stamps.alloc(64);
            (void)hipMemset(p->d_stamps.p, 0, 64 * 8);
        }
        d.stamps = p->d_stamps.p;
    }
}

int plan_create_impl(pya_handle *h, const pya_batch *b, uint32_t flags, const IoReq *io, pya_plan **out) {
    if (!h || !b || !out) return PYA_ERR_ARG;
    *out = nullptr;
    h->err.clear();
    h->err_index = -1;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_config(h);
    if (rc) return rc;
    const uint64_t n = b->n_psm;
    if (n > 0 && (!b->peak_off || !b->pep || !b->pep_off || !b->n_of_mod || !b->max_charge))
        return h->fail(PYA_ERR_ARG, -1, "NULL array in batch");
    if (n >= (1ull << 31)) return h->fail(PYA_ERR_LIMIT, -1, "more than 2^31 PSMs in one batch");
    const bool host_timing = h->kn.host_timing;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pya plan] %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    std::unique_ptr<pya_plan> p(new pya_plan);
    p->h = h;
    for (Bucket &bk : p->buckets) bk.take_knobs(h->kn);      /* (PYA_SB / PYA_GTP / PYA_HASH_PP of THIS handle) */
    p->fusedb.take_knobs(h->kn);
    p->bigloc.take_knobs(h->kn);
    p->flags = flags;
    p->n_psm = n;
    p->peak_off.assign(b->peak_off, b->peak_off + n + 1);
    p->pep_off.assign(b->pep_off, b->pep_off + n + 1);
    p->n_of_mod.assign(b->n_of_mod, b->n_of_mod + n);
    p->max_charge.assign(b->max_charge, b->max_charge + n);
    const bool has_aux = b->aux_off && b->aux_pos && b->aux_mass;
    if (has_aux) p->aux_off.assign(b->aux_off, b->aux_off + n + 1);
    else p->aux_off.assign(n + 1, 0);
    const int64_t peak_base = n ? p->peak_off[0] : 0, pep_base = n ? p->pep_off[0] : 0,
                  aux_base = n ? p->aux_off[0] : 0;
    p->total_peaks = n ? p->peak_off[n] - peak_base : 0;
    const int64_t total_pep = n ? p->pep_off[n] - pep_base : 0;
    const int64_t total_aux = n ? p->aux_off[n] - aux_base : 0;
    if (p->total_peaks < 0 || total_pep < 0 || total_aux < 0)
        return h->fail(PYA_ERR_ARG, -1, "offset arrays are not monotone (the last offset is below the first)");
    p->pep.assign(b->pep + pep_base, b->pep + pep_base + total_pep);
    for (uint64_t i = 0; i <= n; i++) {
        p->peak_off[i] -= peak_base;
        p->pep_off[i] -= pep_base;
        p->aux_off[i] -= aux_base;
    }
    p->ret_off.resize(n + 1);
    {
        int64_t at = 0;                                     /* every PSM's retained table starts at an even entry */
        for (uint64_t i = 0; i < n; i++) {
            p->ret_off[i] = at;
            const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
            at += ((P > 0 ? P : 0) + 1) & ~(int64_t)1;
        }
        p->ret_off[n] = at;
    }
    lap("copy meta");
    p->n_sites.resize(n);
    p->n_sig.resize(n);
    p->order_off.resize(n);
    p->sig_off.resize(n + 1);
    p->ncls.assign(n, 0);
    const uint32_t n_uniq = (uint32_t)h->cfg.n_uniq, n_types = (uint32_t)h->cfg.n_types;
    uint32_t max_P = 1, lut_need = 0, max_k = 1;
    int64_t sig_total = 0;
    /* (tiny batches are launch-bound: the lean instantiation's extra memset + hand-over launch cost
     * more than its occupancy gains there) */
    const bool plain_on = h->cfg.n_nl == 0 && !(flags & PYA_FLAG_KEEP) && !h->kn.no_plain && n >= (uint64_t)h->kn.plain_min;
    /* Pass A (threaded for big batches): the per-letter work -- validate every PSM and count its
     * modifiable residues.  It only finds the first offending PSM; the detailed message comes from
     * the serial checks below, run for that PSM alone. */
    /* fused score + localize kernel: plain settings with one ion type per direction; C(n,k) <= 32 when
     * both directions are scored (one (signature, direction) walker per lane), <= 64 with one */
    const bool both_dirs = h->cfg.n_fwd > 0 && h->cfg.n_fwd < h->cfg.n_types;
    const bool fused_on = plain_on && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 && !h->kn.no_fused;
    /* (both directions and 33 .. 64 site assignments: the fused kernel walks the directions one after the other) */
    const uint32_t fused_max_n = 64u;
    p->fused.assign(n, 0);
    /* score_big.hip: one PSM per 8-wave workgroup, fragment tree shared two levels deep */
    const uint64_t big_min_n = (uint64_t)h->kn.big_min_n;
    const bool big_on = h->cfg.n_nl == 0 && both_dirs && h->cfg.n_fwd == 1 && h->cfg.n_types == 2 && h->mz_error <= 0.49f && !h->kn.no_big;
    p->big.assign(n, 0);
    p->gen.assign(n, 0);
    /* (summary mode with the lean localize route on: the kernel localises what it scores) */
    const bool big_inline_ok = big_on && plain_on && !h->kn.no_big_inline;
    const uint64_t big_inline_max = pya_big_inline_max();
    const bool skip_invalid = (flags & PYA_FLAG_SKIP_INVALID) != 0;
    std::vector<uint8_t> bad(n, 0);
    {
        auto scan = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
                const int64_t L = p->pep_off[i + 1] - p->pep_off[i];
                const int32_t k = p->n_of_mod[i], z = p->max_charge[i];
                bool ok = P > 0 && P <= PYA_MAX_PEAKS && L >= 1 && L <= PYA_MAX_PEPTIDE_LEN && k >= 0 && z >= 1 && z <= PYA_MAX_CHARGE &&
                          p->aux_off[i + 1] >= p->aux_off[i];
                uint32_t ns = 0;
                if (ok) {
                    if (io && io->pre_sites) {
                        ns = io->pre_sites[i];
                        ok = ns != 255u;
                    } else {
                        ok = h->scan_peptide(p->pep.data() + p->pep_off[i], L, &ns);
                    }
                    for (int64_t a = p->aux_off[i]; has_aux && a < p->aux_off[i + 1]; a++)
                        ok = ok && b->aux_pos[aux_base + a] <= (uint32_t)L;
                    ok = ok && ns <= PYA_MAX_SITES;
                }
                bad[i] = ok ? 0 : 1;
                p->n_sites[i] = ok ? (uint8_t)ns : 0;
            }
        };
        unsigned nt = n >= 20000 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (nt == 1) {
            scan(0, n);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(scan, n * t / nt, n * (t + 1) / nt);
            for (auto &x : th) x.join();
        }
    }
    lap("letter scan");
    /* an invalid PSM ends the call with its message -- or, with PYA_FLAG_SKIP_INVALID, is set aside
     * (status PYA_ST_INVALID / PYA_ST_OVER_LIMIT, best_score -1, n_sig -1) while the rest is scored */
    p->pre_status.assign(n, 0);
    uint64_t n_skipped = 0;
    auto reject = [&](int code, uint64_t i, const char *fmt, ...) -> int {
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        if (!skip_invalid) return h->fail(code, (int64_t)i, "%s", buf);
        if (n_skipped == 0) (void)h->fail(code, (int64_t)i, "%s", buf);       /* first message is kept */
        n_skipped++;
        p->pre_status[i] = code == PYA_ERR_LIMIT ? PYA_ST_OVER_LIMIT : PYA_ST_INVALID;
        p->n_sites[i] = 0;
        p->n_sig[i] = 0;
        p->order_off[i] = 0;
        p->sig_off[i] = sig_total;
        p->buckets[0].general_ids.push_back((uint32_t)i);     /* localize writes the "no result" record */
        return PYA_OK;
    };
    for (uint64_t i = 0; i < n; i++) {
        const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
        const int64_t L = p->pep_off[i + 1] - p->pep_off[i];
        const int32_t k = p->n_of_mod[i], z = p->max_charge[i];
        uint32_t ns = p->n_sites[i];
        if (bad[i]) {
            int rc2 = PYA_OK;
            const uint64_t iu = i;
            if (P <= 0) rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: empty spectrum", (unsigned long long)iu);
            else if (P > PYA_MAX_PEAKS)
                rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %lld peaks exceed the limit of %d", (unsigned long long)iu,
                             (long long)P, PYA_MAX_PEAKS);
            else if (L < 1 || L > PYA_MAX_PEPTIDE_LEN)
                rc2 = reject(L < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, i, "PSM %llu: peptide length %lld outside 1..%d",
                             (unsigned long long)iu, (long long)L, PYA_MAX_PEPTIDE_LEN);
            else if (k < 0) rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: negative n_of_mod", (unsigned long long)iu);
            else if (z < 1 || z > PYA_MAX_CHARGE)
                rc2 = reject(z < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, i, "PSM %llu: max_fragment_charge %d outside 1..%d", (unsigned long long)iu, z, PYA_MAX_CHARGE);
            else if (p->aux_off[i + 1] < p->aux_off[i])
                rc2 = reject(PYA_ERR_ARG, i, "PSM %llu: aux_off is not monotone", (unsigned long long)iu);
            else {
                const uint8_t *s = p->pep.data() + p->pep_off[i];
                uint32_t cnt = 0;
                int64_t bad_j = -1;
                for (int64_t j = 0; j < L; j++) {
                    if (!h->is_residue[s[j]] && bad_j < 0) bad_j = j;
                    if (h->letter_modifiable((char)s[j], (size_t)j, (size_t)L)) cnt++;
                }
                int64_t bad_a = -1;
                for (int64_t a = p->aux_off[i]; has_aux && a < p->aux_off[i + 1]; a++)
                    if (b->aux_pos[aux_base + a] > (uint32_t)L && bad_a < 0) bad_a = a;
                if (bad_j >= 0)
                    rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: unknown residue '%c' at position %lld", (unsigned long long)iu,
                                 (char)s[bad_j], (long long)(bad_j + 1));
                else if (bad_a >= 0)
                    rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: aux_mod_pos %u beyond the peptide", (unsigned long long)iu,
                                 b->aux_pos[aux_base + bad_a]);
                else if (cnt > PYA_MAX_SITES)
                    rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %u modifiable residues exceed %d", (unsigned long long)iu, cnt,
                                 PYA_MAX_SITES);
                else rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: rejected by the batch scan", (unsigned long long)iu);
            }
            if (rc2) return rc2;
            continue;
        }
        uint64_t N = 0;
        if ((uint32_t)k <= ns) {
            uint64_t &cached = h->binom_cache[ns][k];
            if (cached == 0) cached = binom(ns, (uint32_t)k);
            N = cached;
        }
        if (N > PYA_MAX_SIGNATURES) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: C(%u,%d) site assignments exceed the limit of %d",
                             (unsigned long long)i, ns, k, PYA_MAX_SIGNATURES);
            if (rc2) return rc2;
            continue;
        }
        const uint32_t per_type = (uint32_t)(L - 1) * (uint32_t)z * n_uniq;
        if (per_type > PYA_MAX_FRAGMENTS_PER_TYPE) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %u fragments per ion type exceed %d", (unsigned long long)i,
                             per_type, PYA_MAX_FRAGMENTS_PER_TYPE);
            if (rc2) return rc2;
            continue;
        }
        if ((uint64_t)per_type * n_types > PYA_MAX_LUT_N) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: up to %llu theoretical fragments per site assignment; the score table covers %d",
                             (unsigned long long)i, (unsigned long long)per_type * n_types, PYA_MAX_LUT_N);
            if (rc2) return rc2;
            continue;
        }
        p->n_sites[i] = (uint8_t)ns;
        p->n_sig[i] = (uint32_t)N;
        uint32_t ooff = 0;
        if (N) {
            uint32_t &co = h->shape_cache[ns][k];
            if (co == 0xffffffffu) co = shape_offset(h, ns, (uint32_t)k);
            ooff = co;
        }
        p->order_off[i] = ooff;
        p->sig_off[i] = sig_total;
        sig_total += (int64_t)N;
        if (P <= PYA_FAST_PEAKS) max_P = std::max<uint32_t>(max_P, (uint32_t)P);    /* (sizes the LDS of the fast kernels) */
        lut_need = std::max(lut_need, per_type * n_types);
        if ((uint32_t)k > max_k) max_k = (uint32_t)k;
        if (P > PYA_FAST_PEAKS || L > PYA_FAST_PEPTIDE_LEN || N > PYA_FAST_SIGNATURES || per_type > PYA_FAST_FRAGMENTS_PER_TYPE || h->all_general()) {
            /* beyond a limit of the fast kernels (or n_top > 10): the general kernel takes the PSM whole */
            p->gen[i] = 1;
            p->gen_ids.push_back((uint32_t)i);
            {
                /* its own slice of the general kernel's scratch: sort area for ITS site assignments, room for ITS competitors */
                const uint32_t push = ((uint32_t)k <= ns ? (uint32_t)k * (ns - (uint32_t)k) : 0u);
                if (p->gen_off.empty()) p->gen_off.push_back(0);
                p->gen_off.push_back(p->gen_off.back() + ((pya_general_scratch_bytes((uint32_t)N, (push + 3u) & ~3u) + 255) & ~(size_t)255));
            }
            p->gen_l_cap = std::max<uint32_t>(p->gen_l_cap, (uint32_t)L);
            p->gen_list_cap = std::max<uint32_t>(p->gen_list_cap, per_type);
            if (P > PYA_FAST_PEAKS) {
                p->bigbin_ids.push_back((uint32_t)i);
                p->bigbin_cap = std::max<uint32_t>(p->bigbin_cap, ((uint32_t)P + 31u) & ~31u);
            }
            continue;
        }
        int cls_of_i = 0;
        if (N > 0 && (uint32_t)k < ns) {
            int bi = 0;
            while (N > kBucketLimits[bi]) bi++;
            cls_of_i = bi;
            /* (its count records hold the cumulative counts as bytes: at most 255 fragments) */
            const uint32_t frags = (both_dirs ? 2u : 1u) * (uint32_t)(L - 1) * (uint32_t)z;
            const bool to_fused = fused_on && N <= fused_max_n && frags <= 255u;
            bool inl = false;
            if (big_on && z == 1 && N > big_min_n && ns >= 11) { /* (its second level shares ten sites: at least eleven) */
                p->big[i] = 1;
                p->big_pos_cap = std::max(p->big_pos_cap, (uint32_t)(L - 1));
                p->big_k_max = std::max(p->big_k_max, (uint32_t)k);
                inl = big_inline_ok && N <= big_inline_max;
            }
            Bucket &bk = to_fused ? p->fusedb : (inl ? p->bigloc : p->buckets[bi]);
            if (inl) p->n_big_inline++;
            if (to_fused) {
                p->fused[i] = 1;
                p->fused_ent_cap = std::max(p->fused_ent_cap, (uint32_t)(L - 1) * (uint32_t)z);
            }
            /* lean localize instantiation: no neutral losses, charge 1, summary mode (it checks the
             * residue masses itself and hands back what it cannot do) */
            if (inl) {
                bk.ids.push_back((uint32_t)i);           /* (p->bigloc: localised by the recounting lean launch) */
            } else if (to_fused || (plain_on && z == 1)) bk.ids.push_back((uint32_t)i);
            else bk.general_ids.push_back((uint32_t)i);
            bk.n_cap = std::max<uint32_t>(bk.n_cap, (uint32_t)N);
            bk.list_cap = std::max<uint32_t>(bk.list_cap, next_pow2(std::max<uint32_t>(per_type, 1)));
            bk.pos_cap = std::max<uint32_t>(bk.pos_cap, (uint32_t)std::max<int64_t>(L - 1, 1));
            bk.n_types = n_types;
            bk.k_max = std::max<uint32_t>(bk.k_max, (uint32_t)k);
            bk.ns_max = std::max<uint32_t>(bk.ns_max, ns);
            bk.push_max = std::max<uint32_t>(bk.push_max, (uint32_t)k * (ns - (uint32_t)k));
            bk.z_max = std::max<uint32_t>(bk.z_max, (uint32_t)z);
            bk.pair_cap = std::max<uint32_t>(bk.pair_cap, (uint32_t)(L - 1) * n_uniq);
            bk.list_max = std::max<uint32_t>(bk.list_max, per_type);
            if (N <= 64) {
                bk.node_words = std::max<uint32_t>(bk.node_words, 2u * (ns + 1u) * (1u + (uint32_t)((N + 7) / 8)));
                auto sc = h->shape_cols.find(ns << 8 | (uint32_t)k);
                if (sc != h->shape_cols.end()) bk.node_cols = std::max(bk.node_cols, sc->second);
            }
        } else {
            Bucket &bk = p->buckets[0];                 /* unambiguous / empty: cheapest launch */
            bk.general_ids.push_back((uint32_t)i);
            bk.n_cap = std::max<uint32_t>(bk.n_cap, (uint32_t)N);
            /* (its peptide still goes through the score kernel of this class, whose residue table is sized by pos_cap) */
            bk.pos_cap = std::max<uint32_t>(bk.pos_cap, (uint32_t)std::max<int64_t>(L - 1, 1));
        }
        p->ncls[i] = (uint8_t)cls_of_i;
    }
    if (p->n_big_inline) {
        const Bucket &bl = p->bigloc;
        p->big_inline = pya_localize_recount_lds_bytes((max_P + 31u) & ~31u, bl.push_cap(), bl.pos_cap, bl.pool_cap(), bl.sb()) <= 64 * 1024;
        if (!p->big_inline) {
            /* (caps that do not fit what score_big's dead tables leave: the separate localize kernels take them) */
            for (uint64_t i = 0; i < n; i++) {
                if (!p->big[i] || p->pre_status[i] || p->n_sig[i] > big_inline_max) continue;
                Bucket &bk = p->buckets[p->ncls[i]];
                bk.ids.push_back((uint32_t)i);                  /* (these are charge-1 PSMs: the lean list) */
                bk.n_cap = std::max(bk.n_cap, p->n_sig[i]);
                bk.list_cap = std::max(bk.list_cap, bl.list_cap);
                bk.pos_cap = std::max(bk.pos_cap, bl.pos_cap);
                bk.n_types = bl.n_types;
                bk.k_max = std::max(bk.k_max, bl.k_max);
                bk.ns_max = std::max(bk.ns_max, bl.ns_max);
                bk.push_max = std::max(bk.push_max, bl.push_max);
                bk.z_max = std::max(bk.z_max, bl.z_max);
            }
            p->n_big_inline = 0;
            p->bigloc.ids.clear();
        }
    }
    for (Bucket &bk : p->buckets) {
        bk.n_plain = (uint32_t)bk.ids.size();
        bk.ids.insert(bk.ids.end(), bk.general_ids.begin(), bk.general_ids.end());
        bk.general_ids.clear();
        bk.general_ids.shrink_to_fit();
    }
    {
        Bucket &fb = p->fusedb;
        fb.n_plain = (uint32_t)fb.ids.size();
        bool keep_fused = !fb.ids.empty();
        if (keep_fused) {
            p->fused_both = both_dirs ? 1u : 0u;
            p->fused_n_cap = (fb.n_cap + 3u) & ~3u;
            p->fused_stride = (both_dirs ? 2u : 1u) * p->fused_n_cap + 4u;
            const uint32_t cap_all = (max_P + 31u) & ~31u;
            keep_fused = pya_fused_lds_bytes(cap_all, p->fused_n_cap, p->fused_stride, fb.pos_cap, p->fused_ent_cap, fb.push_cap(), p->fused_both, 1u) <= 64 * 1024 &&
                         pya_localize_lds_bytes(fb.push_cap(), fb.n_cap, fb.pos_cap, fb.pool_cap(), fb.sb()) <= kMaxLds;
        }
        if (!keep_fused && !fb.ids.empty()) {               /* (huge spectra) back to the two-kernel route */
            Bucket &b0 = p->buckets[0];
            std::vector<uint32_t> lean, general;                /* charge 1 -> lean localize instantiation */
            for (uint32_t id : fb.ids) (p->max_charge[id] == 1 ? lean : general).push_back(id);
            b0.ids.insert(b0.ids.begin(), lean.begin(), lean.end());
            b0.n_plain += (uint32_t)lean.size();
            b0.ids.insert(b0.ids.end(), general.begin(), general.end());
            b0.n_cap = std::max(b0.n_cap, fb.n_cap);
            b0.list_cap = std::max(b0.list_cap, fb.list_cap);
            b0.pos_cap = std::max(b0.pos_cap, fb.pos_cap);
            b0.n_types = std::max(b0.n_types, fb.n_types);
            b0.k_max = std::max(b0.k_max, fb.k_max);
            b0.ns_max = std::max(b0.ns_max, fb.ns_max);
            b0.push_max = std::max(b0.push_max, fb.push_max);
            b0.z_max = std::max(b0.z_max, fb.z_max);
            fb.ids.clear();
            fb.n_plain = 0;
            std::fill(p->fused.begin(), p->fused.end(), 0);
        }
    }
    lap("psm loop");
    p->n_skipped = n_skipped;
    p->sig_off[n] = sig_total;
    p->total_sigs = sig_total;
    p->max_k = max_k;
    p->peak_cap = (max_P + 31u) & ~31u;
    {
        /* peak classes: the median, 90th and 99th percentile and the maximum of the peak counts,
         * rounded up to 32 (one class for small batches) */
        std::vector<uint32_t> caps;
        if (n >= 2048 && !h->kn.one_peak_class) {
            /* (the order statistics from a histogram of the counts -- they are at most PYA_FAST_PEAKS here --: one pass
             * instead of three std::nth_element over the batch) */
            std::vector<uint32_t> hist(PYA_FAST_PEAKS + 2, 0);
            std::vector<uint8_t> global_bin(p->bigbin_ids.empty() ? 0 : n, 0);
            for (uint32_t id : p->bigbin_ids) global_bin[id] = 1;      /* (binned by their own kernel) */
            for (uint64_t i = 0; i < n; i++) {
                uint32_t v = p->pre_status[i] ? 1u : (uint32_t)(p->peak_off[i + 1] - p->peak_off[i]);
                if (!global_bin.empty() && global_bin[i]) v = 1u;
                hist[v > PYA_FAST_PEAKS ? PYA_FAST_PEAKS + 1 : v]++;
            }
            for (double q : {0.5, 0.9, 0.99}) {
                const size_t at = (size_t)(q * (double)(n - 1));
                size_t acc = 0;
                uint32_t v = 0;
                for (; v < hist.size(); v++) {
                    acc += hist[v];
                    if (acc > at) break;
                }
                caps.push_back((v + 31u) & ~31u);
            }
        }
        /* classes only pay when the tail is long: every extra launch has its own ramp-up and tail */
        if (!caps.empty() && p->peak_cap < 2 * caps[0] && !h->kn.peak_classes) caps.clear();
        caps.push_back(p->peak_cap);
        std::sort(caps.begin(), caps.end());
        caps.erase(std::unique(caps.begin(), caps.end()), caps.end());
        const size_t nc = caps.size();
        std::vector<uint32_t> cnt_bin(nc, 0), cnt_score(nc * kNumBuckets, 0), cnt_big(nc, 0);
        std::vector<uint8_t> pcls(n);
        for (uint64_t i = 0; i < n; i++) {
            if (p->pre_status[i]) continue;                  /* set aside: neither binned nor scored */
            const uint32_t P = (uint32_t)(p->peak_off[i + 1] - p->peak_off[i]);
            if (P > PYA_FAST_PEAKS) continue;               /* (pya_bin_global_kernel; scored by the general kernel) */
            size_t c = 0;
            while (caps[c] < P) c++;
            pcls[i] = (uint8_t)c;
            cnt_bin[c]++;
            if (p->fused[i] || p->gen[i]) continue;
            if (p->big[i]) cnt_big[c]++;
            else cnt_score[p->ncls[i] * nc + c]++;
        }
        uint32_t off = 0;
        for (size_t c = 0; c < nc; c++) {
            p->bin_lists.push_back({off, 0u, caps[c], 0u});
            off += cnt_bin[c];
        }
        off = 0;
        for (size_t g = 0; g < nc * kNumBuckets; g++) {
            p->score_lists.push_back({off, 0u, caps[g % nc], (uint32_t)(g / nc)});
            off += cnt_score[g];
        }
        uint32_t n_score = off;
        off = 0;
        for (size_t c = 0; c < nc; c++) {
            p->big_lists.push_back({off, 0u, caps[c], 0u});
            off += cnt_big[c];
        }
        p->big_ids.resize(off);
        {
            uint32_t nb = 0;
            for (size_t c = 0; c < nc; c++) nb += cnt_bin[c];
            p->bin_ids.resize(nb);
        }
        p->score_ids.resize(n_score);
        for (uint64_t i = 0; i < n; i++) {
            if (p->pre_status[i]) continue;
            if (p->peak_off[i + 1] - p->peak_off[i] > PYA_FAST_PEAKS) continue;
            {
                pya_plan::IdList &bl = p->bin_lists[pcls[i]];
                p->bin_ids[bl.off + bl.n++] = (uint32_t)i;
            }
            if (p->fused[i] || p->gen[i]) {
                /* (listed below / in gen_ids) */
            } else if (p->big[i]) {
                pya_plan::IdList &gl = p->big_lists[pcls[i]];
                p->big_ids[gl.off + gl.n++] = (uint32_t)i;
            } else {
                pya_plan::IdList &sl = p->score_lists[p->ncls[i] * nc + pcls[i]];
                p->score_ids[sl.off + sl.n++] = (uint32_t)i;
            }
        }
            {
            const uint32_t ndir = both_dirs ? 2u : 1u;
            p->n_fused_total = 0;
            for (uint64_t i = 0; i < n; i++) p->n_fused_total += (p->fused[i] && !p->pre_status[i]) ? 1u : 0u;
            /* ---- launches of the fused kernel ---- */
            struct Item { uint32_t id, group; size_t need; uint32_t n_cap, pos, ent, push; };
            std::vector<Item> items;
            for (uint64_t i = 0; i < n; i++) {
                if (!p->fused[i] || p->pre_status[i]) continue;
                Item it;
                it.id = (uint32_t)i;
                const uint32_t z = (uint32_t)p->max_charge[i], Lm1 = (uint32_t)(p->pep_off[i + 1] - p->pep_off[i] - 1);
                const uint32_t kk = (uint32_t)p->n_of_mod[i], ns = p->n_sites[i];
                it.group = (uint32_t)pcls[i] * 2 + (z > 1 ? 1u : 0u);
                it.n_cap = (p->n_sig[i] + 3u) & ~3u;
                it.pos = std::max(Lm1, 1u);
                it.ent = std::max(Lm1 * z, 1u);
                it.push = std::min<uint32_t>(PYA_MAX_PUSHED, (kk * (ns - kk) + 7u) & ~7u);
                if (it.push < 8) it.push = 8;
                it.need = pya_fused_lds_bytes(caps[pcls[i]], it.n_cap, ndir * it.n_cap + 4, it.pos, it.ent, it.push, p->fused_both, z > 1 ? 1u : 0u);
                items.push_back(it);
            }
            /* order: (group, LDS need, id).  The items come in id order and (group, need) takes a few hundred values at most, so
             * a stable counting sort over the distinct pairs (kept sorted, found by bisection) does it in O(n log pairs) with
             * eight-byte keys -- r06: std::sort over 100 000 forty-byte items was 3 of the 7 ms of a cfg2 batch's pre-pass */
            {
                std::vector<std::pair<uint64_t, uint32_t>> keys;     /* (group << 40 | need, class id), sorted by key */
                keys.reserve(256);
                std::vector<uint32_t> cls(items.size());
                std::vector<size_t> cnt;
                uint64_t last_key = ~0ull;
                uint32_t last_cls = 0;
                for (size_t t = 0; t < items.size(); t++) {
                    const uint64_t key = ((uint64_t)items[t].group << 40) | (uint64_t)items[t].need;
                    if (key != last_key) {
                        auto it = std::lower_bound(keys.begin(), keys.end(), std::make_pair(key, 0u));
                        if (it == keys.end() || it->first != key) {
                            it = keys.insert(it, std::make_pair(key, (uint32_t)cnt.size()));
                            cnt.push_back(0);
                        }
                        last_key = key;
                        last_cls = it->second;
                    }
                    cls[t] = last_cls;
                    cnt[last_cls]++;
                }
                std::vector<size_t> at(cnt.size());
                size_t acc = 0;
                for (const auto &kc : keys) {                        /* ascending (group, need) */
                    at[kc.second] = acc;
                    acc += cnt[kc.second];
                }
                std::vector<Item> sorted(items.size());
                for (size_t t = 0; t < items.size(); t++) sorted[at[cls[t]]++] = items[t];     /* (stable: ids stay ascending) */
                items.swap(sorted);
            }
            p->fused_ids.resize(items.size());
            size_t g0 = 0;
            while (g0 < items.size()) {
                size_t g1 = g0;
                while (g1 < items.size() && items[g1].group == items[g0].group) g1++;
                /* LDS classes inside the group: cut at the median and the 85th percentile of the footprint when
                 * that buys at least a fifth of the largest footprint (every launch has its own ramp-up and tail) */
                std::vector<size_t> cuts{g0};
                if (g1 - g0 >= 8192 && !h->kn.one_lds_class) {
                    const size_t need_max = items[g1 - 1].need;
                    for (double q : {0.5, 0.85}) {
                        const size_t at = g0 + (size_t)(q * (double)(g1 - g0));
                        size_t cut = at;
                        while (cut < g1 && items[cut].need == items[at].need) cut++;    /* equal footprints stay together */
                        if (cut < g1 && cut > cuts.back() && items[at].need * 5 <= need_max * 4) cuts.push_back(cut);
                    }
                }
                cuts.push_back(g1);
                for (size_t c = 0; c + 1 < cuts.size(); c++) {
                    pya_plan::FusedLaunch fl = {(uint32_t)cuts[c], (uint32_t)(cuts[c + 1] - cuts[c]), caps[items[g0].group / 2],
                                                items[g0].group & 1u, 4, 4, 1, 1, 8};
                    for (size_t t = cuts[c]; t < cuts[c + 1]; t++) {
                        fl.n_cap = std::max(fl.n_cap, items[t].n_cap);
                        fl.pos_cap = std::max(fl.pos_cap, items[t].pos);
                        fl.ent_cap = std::max(fl.ent_cap, items[t].ent);
                        fl.push_cap = std::max(fl.push_cap, items[t].push);
                        p->fused_ids[t] = items[t].id;
                    }
                    fl.stride = ndir * fl.n_cap + 4;            /* + a spare column for lanes without a walker */
                    p->fused_launches.push_back(fl);
                }
                g0 = g1;
            }
        }
    }
    /* packed descriptors: what a kernel needs to know about a PSM before it can fetch anything else,
     * in one cache line (fetched ahead by the fused kernel) */
    p->desc.resize((size_t)n * PYA_DESC_WORDS);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t *w = &p->desc[(size_t)i * PYA_DESC_WORDS];
        const uint64_t L = (uint64_t)std::max<int64_t>(0, std::min<int64_t>(p->pep_off[i + 1] - p->pep_off[i], 0xffff));
        const uint64_t na = (uint64_t)std::max<int64_t>(0, std::min<int64_t>(p->aux_off[i + 1] - p->aux_off[i], 0xffff));
        w[0] = (uint64_t)p->ret_off[i];
        w[1] = (uint64_t)p->pep_off[i];
        w[2] = (uint64_t)p->sig_off[i];
        w[3] = (uint64_t)p->aux_off[i];
        w[4] = L | na << 16 | (uint64_t)((uint32_t)p->n_of_mod[i] & 0xffffu) << 32 | (uint64_t)p->n_sites[i] << 48 |
               (uint64_t)((uint32_t)p->max_charge[i] & 0xffu) << 56;
        w[5] = (uint64_t)p->n_sig[i] | (uint64_t)p->order_off[i] << 32;
    }
    lap("id lists");
    rc = ensure_lut(h, lut_need);
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
        /* (null-stream copies; the kernels that read these tables may run on a non-blocking stream) */
        HIPCHK(h, hipDeviceSynchronize());
        h->order_uploaded = h->order_tab.size();
    }
    if (io && io->max_k < max_k)
        return h->fail(PYA_ERR_ARG, -1, "results.max_k (%u) is smaller than the largest n_of_mod (%u)", io->max_k, max_k);
    for (Bucket &bk : p->buckets) {
        if (bk.ids.empty()) continue;
        size_t need = pya_localize_lds_bytes(bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb());
        if (need > kMaxLds)
            return h->fail(PYA_ERR_LIMIT, (int64_t)bk.ids[0], "LDS budget exceeded (%zu bytes) for the bucket of PSM %u",
                           need, bk.ids[0]);
    }
    if (!p->gen_ids.empty() && pya_general_lds_bytes(p->gen_l_cap, p->gen_list_cap) > kMaxLds)
        return h->fail(PYA_ERR_LIMIT, (int64_t)p->gen_ids[0], "LDS budget exceeded for the general kernel (PSM %u)", p->gen_ids[0]);
    lap("tables");
    /* One device allocation for everything (hipMalloc is ~100 us a call), laid out so that what
     * goes up and what comes back are each one contiguous range. */
    {
        struct Up { size_t off; const void *src; size_t bytes; };
        std::vector<Up> ups;
        size_t total = 0;
        auto reserve = [&](size_t bytes) {
            size_t o = total;
            total += (bytes + 255) & ~(size_t)255;
            return o;
        };
        auto meta = [&](const void *src, size_t bytes) {
            size_t o = reserve(bytes);
            if (src && bytes) ups.push_back({o, src, bytes});
            return o;
        };
        const size_t o_ret_off = meta(p->ret_off.data(), (n + 1) * 8);
        const size_t o_peak_off = meta(p->peak_off.data(), (n + 1) * 8), o_pep_off = meta(p->pep_off.data(), (n + 1) * 8),
                     o_aux_off = meta(p->aux_off.data(), (n + 1) * 8), o_sig_off = meta(p->sig_off.data(), (n + 1) * 8),
                     o_pep = meta(p->pep.data(), p->pep.size()), o_n_sites = meta(p->n_sites.data(), n),
                     o_n_of_mod = meta(p->n_of_mod.data(), n * 4), o_max_charge = meta(p->max_charge.data(), n * 4),
                     o_n_sig = meta(p->n_sig.data(), n * 4), o_order_off = meta(p->order_off.data(), n * 4),
                     o_aux_pos = meta(has_aux ? b->aux_pos + aux_base : nullptr, (size_t)total_aux * 4),
                     o_aux_mass = meta(has_aux ? b->aux_mass + aux_base : nullptr, (size_t)total_aux * 4),
                     o_bin_ids = meta(p->bin_ids.data(), p->bin_ids.size() * 4),
                     o_score_ids = meta(p->score_ids.data(), p->score_ids.size() * 4),
                     o_fused_ids = meta(p->fused_ids.data(), p->fused_ids.size() * 4),
                     o_desc = meta(p->desc.data(), p->desc.size() * 8),
                     o_big_ids = meta(p->big_ids.data(), p->big_ids.size() * 4);
        const size_t o_bigloc_ids = meta(p->bigloc.ids.data(), p->bigloc.ids.size() * 4);
        const size_t o_gen_ids = meta(p->gen_ids.data(), p->gen_ids.size() * 4);
        const size_t o_gen_off = meta(p->gen_off.data(), p->gen_off.size() * 8);
        const size_t o_bigbin_ids = meta(p->bigbin_ids.data(), p->bigbin_ids.size() * 4);
        size_t o_bucket_ids[kNumBuckets];
        for (int i = 0; i < kNumBuckets; i++)
            o_bucket_ids[i] = meta(p->buckets[i].ids.data(), p->buckets[i].ids.size() * 4);
        const bool own_spectra = io && !io->d_mz_ext;
        const size_t o_mz = own_spectra ? meta(io->mz + peak_base, (size_t)p->total_peaks * 8) : 0,
                     o_inten = own_spectra ? meta(io->inten + peak_base, (size_t)p->total_peaks * 8) : 0;
        const size_t h2d_bytes = total;
        p->o_status = reserve(n * 4);
        if (io) {
            const size_t mk = io->max_k;
            p->io_max_k = io->max_k;
            p->o_best_score = reserve(n * 4);
            p->o_best_sig = reserve(n * 8);
            p->o_n_sig_out = reserve(n * 4);
            p->o_ascores = reserve(n * mk * 4);
            p->o_alt = reserve(n * mk * 8);
        }
        p->d2h_bytes = total - p->o_status;
        const size_t o_ret_n = reserve(n * 4),
                     o_ret = reserve((size_t)p->ret_off[n] * sizeof(PeakEntry) + 64),
                     o_grid = reserve(n * PYA_GRID_CELLS * 2), o_redo = reserve((n + 64) * 4), o_redo3 = reserve((2 * n + 64) * 4), o_redo4 = reserve(((size_t)p->n_fused_total + 64) * 4), o_redo5 = reserve(((size_t)p->n_big_inline + 64) * 4), o_ws_top = reserve(n * 16),
                     o_ws = reserve((size_t)sig_total * 4), o_rec = reserve((size_t)sig_total * h->rec_words() * 4),
                     o_sorted = reserve((flags & PYA_FLAG_KEEP) ? (size_t)sig_total * 4 : 0);
        const size_t gen_scratch_bytes = p->gen_off.empty() ? 0 : (size_t)p->gen_off.back();
        const size_t o_gen_scratch = reserve(gen_scratch_bytes);
        p->bigbin_stride = p->bigbin_ids.empty() ? 0 : (pya_bin_global_scratch_bytes(p->bigbin_cap) + 255) & ~(size_t)255;
        const size_t o_bigbin_scratch = reserve(p->bigbin_ids.size() * p->bigbin_stride);
        if (!p->arena.take_if_fits(h->spare_arena, total) && !p->arena.take_if_fits(h->spare_arena2, total))
            HIPCHK(h, p->arena.alloc(total));
        unsigned char *base = p->arena.p;
        p->d_peak_off.adopt(base + o_peak_off, n + 1);
        p->d_pep_off.adopt(base + o_pep_off, n + 1);
        p->d_aux_off.adopt(base + o_aux_off, n + 1);
        p->d_sig_off.adopt(base + o_sig_off, n + 1);
        p->d_pep.adopt(base + o_pep, p->pep.size());
        p->d_n_sites.adopt(base + o_n_sites, n);
        p->d_n_of_mod.adopt(base + o_n_of_mod, n);
        p->d_max_charge.adopt(base + o_max_charge, n);
        p->d_n_sig.adopt(base + o_n_sig, n);
        p->d_order_off.adopt(base + o_order_off, n);
        p->d_aux_pos.adopt(base + o_aux_pos, (size_t)total_aux);
        p->d_aux_mass.adopt(base + o_aux_mass, (size_t)total_aux);
        p->d_bin_ids.adopt(base + o_bin_ids, p->bin_ids.size());
        p->d_score_ids.adopt(base + o_score_ids, p->score_ids.size());
        p->d_fused_ids.adopt(base + o_fused_ids, p->fused_ids.size());
        p->d_redo5.adopt(base + o_redo5, (size_t)p->n_big_inline + 64);
        p->d_desc.adopt(base + o_desc, p->desc.size());
        p->d_big_ids.adopt(base + o_big_ids, p->big_ids.size());
        for (int i = 0; i < kNumBuckets; i++)
            p->buckets[i].d_ids.adopt(base + o_bucket_ids[i], p->buckets[i].ids.size());
        p->bigloc.d_ids.adopt(base + o_bigloc_ids, p->bigloc.ids.size());
        p->d_gen_ids.adopt(base + o_gen_ids, p->gen_ids.size());
        p->d_gen_scratch.adopt(base + o_gen_scratch, gen_scratch_bytes);
        p->d_gen_off.adopt(base + o_gen_off, p->gen_off.size());
        p->d_bigbin_ids.adopt(base + o_bigbin_ids, p->bigbin_ids.size());
        p->d_bigbin_scratch.adopt(base + o_bigbin_scratch, p->bigbin_ids.size() * p->bigbin_stride);
        if (io) {
            if (own_spectra) {
                p->d_mz.adopt(base + o_mz, (size_t)p->total_peaks);
                p->d_inten.adopt(base + o_inten, (size_t)p->total_peaks);
            } else {
                p->d_mz.adopt(io->d_mz_ext, (size_t)p->total_peaks);
                p->d_inten.adopt(io->d_inten_ext, (size_t)p->total_peaks);
            }
            p->d_best_score.adopt(base + p->o_best_score, n);
            p->d_best_sig.adopt(base + p->o_best_sig, n);
            p->d_n_sig_out.adopt(base + p->o_n_sig_out, n);
            p->d_ascores.adopt(base + p->o_ascores, n * io->max_k);
            p->d_alt.adopt(base + p->o_alt, n * io->max_k);
        }
        p->d_status.adopt(base + p->o_status, n);
        p->d_ret_n.adopt(base + o_ret_n, n);
        p->d_ret.adopt(base + o_ret, (size_t)p->ret_off[n] + 8);
        p->d_ret_off.adopt(base + o_ret_off, n + 1);
        p->d_grid.adopt(base + o_grid, n * PYA_GRID_CELLS);
        p->d_redo.adopt(base + o_redo, n + 64);
        p->d_redo3.adopt(base + o_redo3, 2 * n + 64);
        p->d_redo4.adopt(base + o_redo4, (size_t)p->n_fused_total + 64);
        p->d_ws_top.adopt(base + o_ws_top, n * 4);
        p->d_ws.adopt(base + o_ws, (size_t)sig_total);
        p->d_rec.adopt(base + o_rec, (size_t)sig_total * h->rec_words());
        if (flags & PYA_FLAG_KEEP) p->d_sorted.adopt(base + o_sorted, (size_t)sig_total);
        if (h2d_bytes <= kStageLimit) {
            /* small batch: HIP call overhead dominates, so gather on the host and copy once */
            h->stage.resize(std::max(h->stage.size(), h2d_bytes));
            for (const Up &u : ups) std::memcpy(h->stage.data() + u.off, u.src, u.bytes);
            HIPCHK(h, hipMemcpy(base, h->stage.data(), h2d_bytes, hipMemcpyHostToDevice));
            /* (a null-stream copy from pageable memory; the plan may run on a non-blocking stream, which does not wait for it) */
            HIPCHK(h, hipStreamSynchronize(nullptr));
        } else {
            hipStream_t ust = io ? io->stream : nullptr;
            for (const Up &u : ups) HIPCHK(h, hipMemcpyAsync(base + u.off, u.src, u.bytes, hipMemcpyHostToDevice, ust));
            if (ust) HIPCHK(h, hipStreamSynchronize(ust));
            else HIPCHK(h, hipDeviceSynchronize());
        }
    }
    if (n_skipped)      /* bin_spectra never touches these entries, so they keep their code for every run */
    {
        HIPCHK(h, hipMemcpy(p->d_status.p, p->pre_status.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(h, hipStreamSynchronize(nullptr));
    }
    lap("arena+upload");
    {
        /* the fused family beside the others (host_internal.h: pya_plan::fork) when the batch has both */
        bool others = !p->gen_ids.empty();
        for (const pya_plan::IdList &l : p->score_lists) others = others || l.n != 0;
        for (const pya_plan::IdList &l : p->big_lists) others = others || l.n != 0;
        p->fork = !h->kn.no_fork && p->n_fused_total != 0 && others;
        if (p->fork) {
            if (!h->side_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            p->side = h->side_stream;
            HIPCHK(h, hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
    }
    if (flags & PYA_FLAG_TIMING) {
        p->evring.assign(pya_plan::kEvPerRun * pya_plan::kEvRing, nullptr);
        p->evalias.assign(pya_plan::kEvPerRun * pya_plan::kEvRing, 0);
        /* (timing only: nothing synchronises-with the work through these events, so no system-scope fence per record) */
        for (auto &e : p->evring) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    }
    fill_dev(p.get());
    *out = p.release();
    return PYA_OK;
}

int pya_plan_create(pya_handle *h, const pya_batch *b, uint32_t flags, pya_plan **out) {
    return plan_create_impl(h, b, flags, nullptr, out);
}

int pya_plan_run(pya_plan *p, const double *d_mz, const double *d_inten, void *hip_stream,
                 const pya_results *o) {
    if (!p || !o) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (p->n_psm == 0) return PYA_OK;
    if (!d_mz || !d_inten || !o->best_score || !o->best_sig || !o->n_sig || !o->ascores || !o->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL device pointer passed to pya_plan_run");
    if (o->max_k < p->max_k)
        return h->fail(PYA_ERR_ARG, -1, "results.max_k (%u) is smaller than the largest n_of_mod (%u)",
                       o->max_k, p->max_k);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    refresh_shared(p);
    BatchDev d = p->dev;
    d.mz = d_mz;
    d.inten = d_inten;
    d.best_score = o->best_score;
    d.best_sig = o->best_sig;
    d.n_sig_out = o->n_sig;
    d.ascores = o->ascores;
    d.alt_mask = o->alt_mask;
    d.max_k = o->max_k;
    const bool timing = p->flags & PYA_FLAG_TIMING;
    /* a handful of PSMs (PyAscore.score is a batch of one) is launch-bound: one fused launch, one
     * wavefront per PSM, instead of the five of the three-kernel path (tiny_batch.hip) */
    const uint64_t tiny_max = (uint64_t)h->kn.tiny_max;
    bool tiny = !timing && p->n_psm <= tiny_max && p->n_skipped == 0 && !h->kn.no_tiny && p->gen_ids.empty();
    Bucket m;                                               /* caps that cover every PSM of the batch */
    m.take_knobs(h->kn);
    uint32_t prefix = 0, compact = 0;
    if (tiny) {
        /* (the PSMs the fused kernel would take are accounted in their own bucket: its caps count too --
         * leaving them out sized this launch's work areas for the other PSMs only) */
        std::vector<const Bucket *> all;
        for (const Bucket &bk : p->buckets) all.push_back(&bk);
        all.push_back(&p->fusedb);
        all.push_back(&p->bigloc);
        for (const Bucket *pbk : all) {
            const Bucket &bk = *pbk;
            if (bk.ids.empty() && pbk != &p->bigloc) continue;
            if (pbk == &p->bigloc && p->n_big_inline == 0) continue;
            m.n_cap = std::max(m.n_cap, bk.n_cap);
            m.list_cap = std::max(m.list_cap, bk.list_cap);
            m.pos_cap = std::max(m.pos_cap, bk.pos_cap);
            m.n_types = std::max(m.n_types, bk.n_types);
            m.k_max = std::max(m.k_max, bk.k_max);
            m.push_max = std::max(m.push_max, bk.push_max);
            m.z_max = std::max(m.z_max, bk.z_max);
        }
        prefix = (m.n_cap >= 128 && !h->kn.no_prefix) ? 1u : 0u;
        compact = (h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 && m.z_max == 1) ? 1u : 0u;
        /* the merged caps (maxima over the buckets) can ask for more LDS than any single bucket does:
         * such a batch takes the three-kernel path, whose launches are sized per bucket */
        if (pya_tiny_lds_bytes(p->peak_cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, m.push_cap(), m.n_cap, m.pos_cap,
                               m.pool_cap(), m.sb()) > kMaxLds)
            tiny = false;
    }
    if (tiny) {
        int e = pya_launch_tiny(&d, (uint32_t)p->n_psm, p->peak_cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact,
                                m.push_cap(), m.n_cap, m.pos_cap, m.pool_cap(), m.sb(), m.gtp(), st);
        if (e) return h->hip_fail((hipError_t)e, "tiny_batch launch");
        p->last_stream = st;
        p->ran = true;
        p->dev = d;
        return PYA_OK;
    }
    hipEvent_t *ev = timing ? p->ev_set(p->ev_runs) : nullptr;
    uint8_t *alias = timing ? p->ev_alias(p->ev_runs) : nullptr;
    /* boundary i of the run: a new event if the family before it launched anything, else the previous boundary's */
    auto mark = [&](int i, bool launched) -> int {
        if (!timing) return 0;
        if (i > 0 && !launched) {
            alias[i] = alias[i - 1];
            return 0;
        }
        alias[i] = (uint8_t)i;
        return (int)hipEventRecord(ev[i], st);
    };
    if (mark(0, true)) return h->hip_fail(hipGetLastError(), "hipEventRecord");
    /* hand-over counts (bin_spectra's, the fused kernel's, the recount's): two sets at the head of d_redo, taken in turn;
     * the binning kernel of a run zeroes the set of the next (bin_spectra.hip), so only a plan without such a launch --
     * and the first run -- needs a memset */
    uint32_t *cnt = p->d_redo.p + 8 * (p->n_runs & 1u);
    d.redo_count = cnt;
    d.redo4_count = cnt + 1;
    d.zero_next = p->d_redo.p + 8 * ((p->n_runs + 1u) & 1u);
    /* (the binning kernel zeroes the next run's set only when it is launched: a plan whose lists are all empty -- every
     * spectrum binned by the global kernel, or set aside -- has nobody to do it) */
    bool any_bin = false;
    for (const pya_plan::IdList &l : p->bin_lists) any_bin = any_bin || l.n != 0;
    if (p->n_runs == 0 || !any_bin) HIPCHK(h, hipMemsetAsync(p->d_redo.p, 0, 16 * sizeof(uint32_t), st));
    p->n_runs++;
    int e = 0;
    for (const pya_plan::IdList &l : p->bin_lists) {
        /* dense classes: selection first (bin_select.hip.h); the all-pairs ranking of bin_fast is O(window^2) and holds 13 bytes
         * of LDS per raw peak */
        const uint32_t scap = (uint32_t)std::min<int64_t>(std::max<int64_t>(h->kn.bin_select_scap, 64), 4096) & ~31u;
        if ((int64_t)l.cap > h->kn.bin_select_min)
            e = pya_launch_bin_select(&d, p->d_bin_ids.p + l.off, l.n, scap, st);
        else
            e = pya_launch_bin(&d, p->d_bin_ids.p + l.off, l.n, l.cap, st);
        if (e) return h->hip_fail((hipError_t)e, "bin_spectra launch");
    }
    e = pya_launch_bin_exact(&d, (uint32_t)p->n_psm, p->peak_cap, st);
    if (e) return h->hip_fail((hipError_t)e, "bin_spectra (exact) launch");
    e = pya_launch_bin_global(&d, p->d_bigbin_ids.p, (uint32_t)p->bigbin_ids.size(), p->d_bigbin_scratch.p, p->bigbin_stride, p->bigbin_cap, st);
    if (e) return h->hip_fail((hipError_t)e, "bin_spectra (global) launch");
    if (mark(1, true)) return h->hip_fail(hipGetLastError(), "hipEventRecord");
    /* the fused family: few site assignments, plain settings, scored and localised in one pass (score_localize.hip), one PSM
     * per wavefront; what it hands over goes through the general localize instantiation.  On `fs`: the plan's side stream when
     * the run forks (everything below is independent of it: other PSMs, other hand-over lists), else the caller's. */
    auto fused_family = [&](hipStream_t fs) -> int {
        const Bucket &fb = p->fusedb;
        for (const pya_plan::FusedLaunch &l : p->fused_launches) {
            int fe = pya_launch_fused(&d, p->d_fused_ids.p + l.off, l.n, l.cap, l.n_cap, l.stride, l.pos_cap, l.ent_cap, l.push_cap,
                                      p->fused_both, l.multi_z, d.redo4_count, d.redo4_ids, fs);
            if (fe) return h->hip_fail((hipError_t)fe, "score_localize launch");
        }
        int fe = pya_launch_localize_redo(&d, d.redo4_count, d.redo4_ids, p->n_fused_total, fb.push_cap(), fb.n_cap,
                                          fb.pos_cap, fb.pool_cap(), fb.sb(), fb.gtp(), fs);
        if (fe) return h->hip_fail((hipError_t)fe, "localize (hand-over) launch");
        return 0;
    };
    if (timing) alias[5] = 0;
    if (p->fork) {
        HIPCHK(h, hipEventRecord(p->ev_fork, st));
        HIPCHK(h, hipStreamWaitEvent(p->side, p->ev_fork, 0));
        if (timing) {
            alias[5] = 1;
            HIPCHK(h, hipEventRecord(ev[5], p->side));
        }
        const int fe = fused_family(p->side);
        if (fe) return fe;
        if (timing) HIPCHK(h, hipEventRecord(ev[6], p->side));
        HIPCHK(h, hipEventRecord(p->ev_join, p->side));
    }
    for (const pya_plan::IdList &l : p->score_lists) {
        /* classes with C(n,k) > 64 share the walk over the first sites between signatures */
        const uint32_t prefix = (p->buckets[l.ncls].n_cap >= 128 && !h->kn.no_prefix) ? 1u : 0u;
        /* every PSM of the class on the straight-line walker: compact prefix entries */
        const uint32_t compact = (h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 &&
                                  p->buckets[l.ncls].z_max == 1) ? 1u : 0u;
        /* general settings (neutral losses, several ion types per direction): one lookup set per distinct node of
         * the assignment tree instead of one per signature (score_core.hip.h: score_nodes_dir) */
        const Bucket &sbk = p->buckets[l.ncls];
        const bool general = h->cfg.n_nl != 0 || h->cfg.n_fwd > 1 || h->cfg.n_types - h->cfg.n_fwd > 1;
        /* many site assignments (classes above 64) under the plain settings with both directions, charge 1: the count-node
         * table decides the fragments (score_cnt.hip) */
        if (l.ncls >= 1 && !general && h->cfg.n_fwd == 1 && h->cfg.n_types == 2 && sbk.z_max == 1 && h->mz_error <= 0.49f &&
            sbk.k_max + 1u <= 31u && sbk.ns_max <= 32u && !h->kn.no_cnt && !(h->kn.debug & 0x8000u)) {
            uint32_t kc = 8u;
            while (kc < sbk.k_max + 1u) kc <<= 1;
            if (pya_score_cnt_lds_bytes(l.cap, sbk.pos_cap, kc, sbk.k_max, sbk.ns_max) <= 64u * 1024u) {
                e = pya_launch_score_cnt(&d, p->d_score_ids.p + l.off, l.n, l.cap, sbk.pos_cap, kc, sbk.k_max, sbk.ns_max, st);
                if (e) return h->hip_fail((hipError_t)e, "score_cnt launch");
                continue;
            }
        }
        /* general settings: the count nodes carry over when the loss variants depend on the count too (score_cntg.hip; the
         * kernel checks that per peptide and walks what does not qualify) */
        if (general && h->mz_error <= 0.49f && sbk.k_max + 1u <= 31u && sbk.ns_max <= 32u && !h->kn.no_cnt && !(h->kn.debug & 0x8000u)) {
            const uint32_t nnl_g = (uint32_t)h->cfg.n_nl, nl_cap_g = nnl_g >= 4u ? 256u : (nnl_g == 0u ? 4u : 1u << (2u * nnl_g));
            if (pya_score_cntg_lds_bytes(l.cap, sbk.pos_cap, sbk.k_max, sbk.ns_max, nl_cap_g) <= 64u * 1024u) {
                e = pya_launch_score_cntg(&d, p->d_score_ids.p + l.off, l.n, l.cap, sbk.pos_cap, sbk.k_max, sbk.ns_max, nl_cap_g, st);
                if (e) return h->hip_fail((hipError_t)e, "score_cntg launch");
                continue;
            }
        }
        uint32_t node_cap = 0, node_cols = std::max<uint32_t>(8u, (sbk.node_cols + 7u) & ~7u);
        /* (the node kernel's LDS decides its occupancy: residue and loss-state tables by the launch, room for 320 nodes
         * per direction -- cfg4's shape needs 186 on average, 328 at most; a direction with more is walked) */
        const uint32_t res_cap = std::min<uint32_t>(64u, (sbk.pos_cap + 1u + 3u) & ~3u);
        const uint32_t nnl_s = (uint32_t)h->cfg.n_nl, nl_cap = nnl_s >= 4u ? 256u : (nnl_s == 0u ? 4u : 1u << (2u * nnl_s));
        if (general && !prefix && !h->kn.no_nodes && sbk.node_words) {
            node_cap = std::min<uint32_t>(320u, (sbk.pos_cap * std::min<uint32_t>(sbk.n_cap, 64u) + 1u) & ~1u);
            if (h->kn.node_cap >= 0) node_cap = (uint32_t)h->kn.node_cap & ~1u;
            if (pya_score_node_lds_bytes(l.cap, h->cfg.n_nl != 0 ? 1u : 0u, node_cap, node_cols, sbk.node_words, res_cap, nl_cap) > 64u * 1024u)
                node_cap = 0;
        }
        e = pya_launch_score(&d, p->d_score_ids.p + l.off, l.n, l.cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, node_cap, node_cols,
                             sbk.node_words, res_cap, nl_cap, st);
        if (e) return h->hip_fail((hipError_t)e, "score_signatures launch");
    }
    for (const pya_plan::IdList &l : p->big_lists) {
        e = pya_launch_score_big(&d, p->d_big_ids.p + l.off, l.n, l.cap, p->big_pos_cap, p->big_kc(), p->big_inline ? 1u : 0u, st);
        if (e) return h->hip_fail((hipError_t)e, "score_big launch");
    }
    {
        bool any = false;                                    /* (lists exist per class even when no PSM is in them) */
        for (const pya_plan::IdList &l : p->score_lists) any = any || l.n != 0;
        for (const pya_plan::IdList &l : p->big_lists) any = any || l.n != 0;
        if (mark(2, any)) return h->hip_fail(hipGetLastError(), "hipEventRecord");
    }
    if (p->n_fused_total && !p->fork) {
        e = fused_family(st);
        if (e) return e;
    }
    if (mark(3, p->n_fused_total != 0 && !p->fork)) return h->hip_fail(hipGetLastError(), "hipEventRecord");
    if (p->big_inline && !p->bigloc.ids.empty()) {
        /* what score_big scored in its summary mode: the lean body with recounted signatures and the winner score_big
         * named; what that declines is scored again with count records and goes to the general localize body */
        const Bucket &bl = p->bigloc;
        e = pya_launch_localize_recount(&d, bl.d_ids.p, (uint32_t)bl.ids.size(), 0u, bl.push_cap(), bl.pos_cap, bl.pool_cap(),
                                        bl.sb(), bl.gtp(), cnt + 2, p->d_redo5.p + 64, st);
        if (e) return h->hip_fail((hipError_t)e, "localize (recount) launch");
        e = pya_launch_score_big_list(&d, cnt + 2, p->d_redo5.p + 64, p->n_big_inline, p->peak_cap, p->big_pos_cap, p->big_kc(), st);
        if (e) return h->hip_fail((hipError_t)e, "score_big (hand-over) launch");
        e = pya_launch_localize_redo(&d, cnt + 2, p->d_redo5.p + 64, p->n_big_inline, bl.push_cap(), (uint32_t)pya_big_inline_max(),
                                     bl.pos_cap, bl.pool_cap(), bl.sb(), bl.gtp(), st);
        if (e) return h->hip_fail((hipError_t)e, "localize (score_big hand-over) launch");
    }
    for (Bucket &bk : p->buckets) {
        /* more than sort_room_max signatures: the lean launch without room for the sort emulation (LDS ->
         * occupancy); PSMs with a tie at the top go through the hand-over list to a second lean pass that has it */
        const uint32_t sort_room = (bk.n_cap <= h->kn.sort_room_max || h->kn.sort_room) ? 1u : 0u;
        e = pya_launch_localize(&d, bk.d_ids.p, bk.n_plain, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb(),
                                bk.gtp(), 1u, sort_room, st);
        if (e) return h->hip_fail((hipError_t)e, "localize launch");
        const uint32_t nnl = (uint32_t)h->cfg.n_nl;
        if (h->kn.host_timing && !bk.ids.empty() && p->n_runs == 1)    /* diagnostics: what decides the general route's occupancy */
            std::fprintf(stderr, "[pya plan] localize bucket: %zu PSMs, LDS hash route %zu B (vc %u hs %u pp %u sb %u push %u), list route %zu B\n",
                         bk.ids.size(), pya_localize_hash_lds_bytes(bk.push_cap(), bk.n_cap, bk.pos_cap, bk.sb(), bk.hash_vc(), bk.hash_hs(),
                                                                    bk.hash_pp(), p->max_k, nnl),
                         bk.hash_vc(), bk.hash_hs(), bk.hash_pp(), bk.sb(), bk.push_cap(),
                         pya_localize_lds_bytes(bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb()));
        if (!h->kn.no_loc_hash && bk.hash_ok(p->max_k, nnl))
            e = pya_launch_localize_hash(&d, bk.d_ids.p + bk.n_plain, (uint32_t)bk.ids.size() - bk.n_plain, bk.push_cap(), bk.n_cap,
                                         bk.pos_cap, bk.pool_cap(), bk.sb(), bk.gtp(), bk.hash_vc(), bk.hash_hs(), bk.hash_pp(),
                                         nnl, st);
        else
        e = pya_launch_localize(&d, bk.d_ids.p + bk.n_plain, (uint32_t)bk.ids.size() - bk.n_plain, bk.push_cap(), bk.n_cap,
                                bk.pos_cap, bk.pool_cap(), bk.sb(), bk.gtp(), 0u, 1u, st);
        if (e) return h->hip_fail((hipError_t)e, "localize launch");
    }
    if (!p->gen_ids.empty()) {
        e = pya_launch_general(&d, p->d_gen_ids.p, (uint32_t)p->gen_ids.size(), p->d_gen_scratch.p, p->d_gen_off.p, p->gen_l_cap,
                               p->gen_list_cap, st);
        if (e) return h->hip_fail((hipError_t)e, "general kernel launch");
    }
    if (timing) {
        bool loc = (p->big_inline && !p->bigloc.ids.empty()) || !p->gen_ids.empty();
        for (const Bucket &bk : p->buckets) loc = loc || !bk.ids.empty();
        if (mark(4, loc)) return h->hip_fail(hipGetLastError(), "hipEventRecord");
        p->ev_runs++;
        if (p->ev_runs - p->ev_read > pya_plan::kEvRing) p->ev_read = p->ev_runs - pya_plan::kEvRing;   /* (overwritten) */
    }
    /* the join: whatever the caller enqueues behind this run waits for the side stream too (after boundary 4, so that the
     * localize family's interval does not include the wait) */
    if (p->fork) HIPCHK(h, hipStreamWaitEvent(st, p->ev_join, 0));
    p->last_stream = st;
    p->ran = true;
    p->dev = d;
    return PYA_OK;
}

int pya_plan_timings(pya_plan *p, float ms[4]) {
    if (!p || !ms) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (!(p->flags & PYA_FLAG_TIMING) || p->ev_runs == 0) return h->fail(PYA_ERR_STATE, -1, "plan has no timing events");
    hipEvent_t *ev = p->ev_set(p->ev_runs - 1);
    const uint8_t *al = p->ev_alias(p->ev_runs - 1);
    HIPCHK(h, hipEventSynchronize(ev[al[4]]));
    if (al[5]) HIPCHK(h, hipEventSynchronize(ev[6]));
    for (int i = 0; i < 4; i++) {
        ms[i] = 0.f;
        if (al[i] != al[i + 1]) HIPCHK(h, hipEventElapsedTime(&ms[i], ev[al[i]], ev[al[i + 1]]));
    }
    if (al[5]) HIPCHK(h, hipEventElapsedTime(&ms[2], ev[5], ev[6]));      /* (the fused family on the side stream) */
    return PYA_OK;
}

int pya_plan_timings_sum(pya_plan *p, double ms[4], uint32_t *n_runs) {
    if (!p || !ms || !n_runs) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (!(p->flags & PYA_FLAG_TIMING)) return h->fail(PYA_ERR_STATE, -1, "plan has no timing events");
    for (int i = 0; i < 4; i++) ms[i] = 0.;
    *n_runs = (uint32_t)(p->ev_runs - p->ev_read);
    if (*n_runs == 0) return PYA_OK;
    HIPCHK(h, hipEventSynchronize(p->ev_set(p->ev_runs - 1)[p->ev_alias(p->ev_runs - 1)[4]]));
    if (p->ev_alias(p->ev_runs - 1)[5]) HIPCHK(h, hipEventSynchronize(p->ev_set(p->ev_runs - 1)[6]));
    for (uint64_t r = p->ev_read; r < p->ev_runs; r++) {
        hipEvent_t *ev = p->ev_set(r);
        const uint8_t *al = p->ev_alias(r);
        for (int i = 0; i < 4; i++) {
            float t = 0.f;
            if (al[i] != al[i + 1]) HIPCHK(h, hipEventElapsedTime(&t, ev[al[i]], ev[al[i + 1]]));
            ms[i] += (double)t;
        }
        if (al[5]) {                                          /* (the fused family on the side stream) */
            float t = 0.f;
            HIPCHK(h, hipEventElapsedTime(&t, ev[5], ev[6]));
            ms[2] += (double)t;
        }
    }
    p->ev_read = p->ev_runs;
    return PYA_OK;
}

int check_status(pya_handle *h, const int32_t *st, uint64_t n, bool skip_invalid) {
    if (skip_invalid) return PYA_OK;                    /* codes are reported per PSM instead */
    for (uint64_t i = 0; i < n; i++) {
        switch (st[i]) {
            case PYA_ST_OK: break;
            case PYA_ST_INVALID:
            case PYA_ST_OVER_LIMIT:
                return h->fail(st[i] == PYA_ST_INVALID ? PYA_ERR_PSM : PYA_ERR_LIMIT, (int64_t)i,
                               "PSM %llu was set aside by the host pre-pass", (unsigned long long)i);
            case PYA_ST_NO_BINS:
                return h->fail(PYA_ERR_PSM, (int64_t)i, "PSM %llu: all peaks sit on one multiple of 100 m/z; the "
                               "spectrum has no windows", (unsigned long long)i);
            case PYA_ST_TOO_MANY_BINS:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: more than 65535 m/z windows", (unsigned long long)i);
            case PYA_ST_LUT_RANGE:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: trial count outside the score table", (unsigned long long)i);
            case PYA_ST_PUSHED_OVERFLOW:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: more than %d tied competitors", (unsigned long long)i, PYA_MAX_PUSHED);
            case PYA_ST_ROUTE_CAPS:
                return h->fail(PYA_ERR_STATE, (int64_t)i, "PSM %llu reached a kernel whose launch was not sized for it (modifications or "
                               "modifiable residues beyond the launch's caps): a routing error of this library", (unsigned long long)i);
            default:
                return h->fail(PYA_ERR_HIP, (int64_t)i, "PSM %llu: unexpected kernel status %d", (unsigned long long)i, st[i]);
        }
    }
    return PYA_OK;
}

int pya_plan_check(pya_plan *p) {
    if (!p) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (!p->ran || p->n_psm == 0) return PYA_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(p->last_stream));
    std::vector<int32_t> st(p->n_psm);
    HIPCHK(h, hipMemcpy(st.data(), p->d_status.p, p->n_psm * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h->kn.host_timing) {                               /* diagnostics: how many PSMs the lean kernels handed over */
        uint32_t r3 = 0, r4 = 0;
        (void)hipMemcpy(&r3, p->d_redo3.p, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&r4, p->d_redo.p + 8 * ((p->n_runs + 1u) & 1u) + 1, 4, hipMemcpyDeviceToHost);   /* (the last run's set) */
        std::fprintf(stderr, "[pya plan] handed over: %u by the lean localize instantiation (last bucket), %u of %u by the fused kernel\n",
                     r3, r4, p->n_fused_total);
    }
    const bool skip = (p->flags & PYA_FLAG_SKIP_INVALID) != 0;
    if (skip) h->last_status = st;
    return check_status(h, st.data(), p->n_psm, skip);
}

int pya_pack_records(pya_handle *h, const pya_results *d_res, uint64_t n_psm, uint32_t k, int32_t *d_out, void *hip_stream) {
    if (!h) return PYA_ERR_ARG;
    if (!d_res || !d_out || !d_res->best_score || !d_res->best_sig || !d_res->n_sig || !d_res->ascores || !d_res->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "pya_pack_records: null result array");
    if (k < d_res->max_k || k > 64 || d_res->max_k == 0)
        return h->fail(PYA_ERR_ARG, -1, "pya_pack_records: record width k = %u is narrower than the results' rows (%u) or above 64", k,
                       d_res->max_k);
    HIPCHK(h, hipSetDevice(h->device));
    const int e = pya_launch_pack_records(d_res->best_score, d_res->n_sig, d_res->best_sig, d_res->ascores, d_res->alt_mask, k,
                                          d_res->max_k, n_psm, d_out, (hipStream_t)hip_stream);
    if (e) return h->fail(PYA_ERR_HIP, -1, "pya_pack_records: launch failed (%d)", e);
    return PYA_OK;
}

uint64_t pya_plan_workspace_bytes(const pya_plan *p) { return p ? p->workspace_bytes() : 0; }
uint64_t pya_plan_total_signatures(const pya_plan *p) { return p ? (uint64_t)p->total_sigs : 0; }

void pya_plan_destroy(pya_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->h->device);
    if (p->d_stamps.p) {
        unsigned long long v[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(v, p->d_stamps.p, sizeof v, hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int i = 0; i < 64; i++) tot += v[i];
        std::fprintf(stderr, "[pya stamps] total %llu\n", tot);
        for (int i = 0; i < 64; i++)
            if (v[i]) std::fprintf(stderr, "[pya stamps] phase %2d: %12llu  %5.1f%%\n", i, v[i], 100.0 * v[i] / tot);
    }
    if (p->h->kept == p) p->h->kept = nullptr;
    if (p->h->one.view == p) p->h->one.view = nullptr;       /* (pya_score_one's retained view of its workspace) */
    if (!p->quiesced) (void)hipDeviceSynchronize();     /* nothing may still be using the buffers */
    if (!p->h->spare_arena.p) p->arena.give_to(p->h->spare_arena);
    else if (!p->h->spare_arena2.p) p->arena.give_to(p->h->spare_arena2);
    else if (p->h->spare_arena.n <= p->h->spare_arena2.n) p->arena.give_to(p->h->spare_arena);
    else p->arena.give_to(p->h->spare_arena2);
    delete p;
}
