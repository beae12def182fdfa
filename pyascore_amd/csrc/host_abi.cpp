/* host_abi.cpp -- the C ABI of include/pyascore_hip.h that is not a plan, a batch call or the one-PSM path: handles, settings,
 * strings, retained records, ambiguity. */
#include "host_internal.h"
#include "../../include/pyascore_debug.h"

extern "C" {


int pya_create(const pya_config *cfg, pya_handle **out) {
    if (!cfg || !out) return PYA_ERR_ARG;
    *out = nullptr;
    std::unique_ptr<pya_handle> h(new pya_handle);
    *out = h.get();                                    /* so the caller can read the message */
    pya_handle *hp = h.release();
    if (!cfg->mod_group || !cfg->fragment_types) return hp->fail(PYA_ERR_ARG, -1, "NULL string in config");
    /* n_top peaks retained per window = depths scored.  Below 10 the reference's weighted sum reads past its scores
     * (cpp/Ascore.cpp:135-137: undefined); above 10 it retains, counts and scores n_top depths, weights the first ten
     * and searches all of them for the depth of an Ascore (:15-36, :123-139, :164-172) -- the general kernel does that,
     * for every PSM of such a scorer. */
    if (cfg->n_top < PYA_NTOP || cfg->n_top > PYA_NTOP_MAX)
        return hp->fail(PYA_ERR_ARG, -1, "n_top must be %d..%d (the PepScore weights are %d long, Ascore.cpp:16-18; "
                        "below that the reference reads past its scores); got %u", PYA_NTOP, PYA_NTOP_MAX, PYA_NTOP, cfg->n_top);
    hp->n_top = cfg->n_top;
    if (!(cfg->bin_size > 0.f)) return hp->fail(PYA_ERR_ARG, -1, "bin_size must be positive");
    if (!(cfg->mz_error > 0.f) || !(cfg->mz_error < 50.f))
        return hp->fail(PYA_ERR_ARG, -1, "mz_error must be in (0, 50)");
    std::string ft = cfg->fragment_types;
    if (ft.empty() || ft.size() > PYA_MAX_FRAGMENT_TYPES)
        return hp->fail(PYA_ERR_ARG, -1, "fragment_types must name 1..%d ion types", PYA_MAX_FRAGMENT_TYPES);
    for (char t : ft)
        if (!is_forward(t) && !is_backward(t))
            return hp->fail(PYA_ERR_ARG, -1, "unknown fragment type '%c' (b, c, y, z, Z are supported)", t);
    hp->device = cfg->device;
    hp->bin_size = cfg->bin_size;
    hp->mod_mass = cfg->mod_mass;
    hp->mz_error = cfg->mz_error;
    hp->mod_group = cfg->mod_group;
    hp->fragment_types = ft;
    hp->build_letter_tables();
    read_knobs(hp->kn);
    std::memset(hp->shape_cache, 0xff, sizeof hp->shape_cache);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return hp->fail(PYA_ERR_HIP, -1, "no HIP device available (%s); this library has no CPU path",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return hp->fail(PYA_ERR_ARG, -1, "device %d out of range", cfg->device);
    HIPCHK(hp, hipSetDevice(cfg->device));
    int rc = build_dev_config(hp);
    if (rc) return rc;
    return PYA_OK;
}

void pya_destroy(pya_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->kept) pya_plan_destroy(h->kept);                 /* (unhooks the one-PSM view if that is what it is) */
    if (h->one.view) delete h->one.view;
    if (h->one.host) (void)hipHostFree(h->one.host);
    if (h->one.stream) (void)hipStreamDestroy(h->one.stream);
    for (void *ps : h->pinned_stage)
        if (ps) (void)hipHostFree(ps);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->run_stream) (void)hipStreamDestroy(h->run_stream);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    delete h;
}

int pya_reload_env(pya_handle *h) {
    if (!h) return PYA_ERR_ARG;
    read_knobs(h->kn);
    h->one.have_last = false;        /* (a replay of the last pya_score_one PSM would run under other switches) */
    return PYA_OK;
}

int pya_set_debug(pya_handle *h, const char *key, const char *value) {
    if (!h || !key) return PYA_ERR_ARG;
    if (!set_knob(h->kn, key, value)) return h->fail(PYA_ERR_ARG, -1, "pya_set_debug: \"%s\" is not a debug switch", key);
    h->one.have_last = false;        /* (a replay of the last pya_score_one PSM would run under other switches) */
    return PYA_OK;
}

const char *pya_last_error(const pya_handle *h) { return h ? h->err.c_str() : "NULL handle"; }
int64_t pya_error_index(const pya_handle *h) { return h ? h->err_index : -1; }

int pya_add_neutral_loss(pya_handle *h, const char *group, float mass) {
    if (!h || !group) return PYA_ERR_ARG;
    std::map<char, float> saved = h->nl;
    for (const char *c = group; *c; c++) h->nl[*c] = mass;     /* ModifiedPeptide.cpp:99-103 */
    h->cfg_dirty = true;
    h->one.have_last = false;        /* (a replay of the last pya_score_one PSM would score it under the new settings) */
    int rc = build_dev_config(h);
    if (rc) {
        h->nl = saved;
        build_dev_config(h);
    }
    return rc;
}

int pya_count_sites(const pya_handle *h, const uint8_t *pep, uint64_t L, int32_t *n_sites, uint16_t *site_pos) {
    if (!h || !pep || !n_sites) return PYA_ERR_ARG;
    int n = 0;
    for (uint64_t i = 0; i < L; i++)
        if (h->letter_modifiable((char)pep[i], i, L)) {
            if (site_pos && n < PYA_MAX_PEPTIDE_LEN) site_pos[n] = (uint16_t)i;
            n++;
        }
    *n_sites = n;
    return PYA_OK;
}

/* ModifiedPeptide::getPeptide, cpp/ModifiedPeptide.cpp:199-253 */
static void format_one(const pya_handle *h, const uint8_t *pep, uint64_t L, int32_t n_of_mod, const uint32_t *aux_pos,
                       const float *aux_mass, uint64_t n_aux, uint64_t sig_bits, int32_t sig_len, std::string &out) {
    size_t sites[PYA_MAX_PEPTIDE_LEN];
    size_t n = 0;
    for (uint64_t i = 0; i < L; i++)
        if (h->letter_modifiable((char)pep[i], i, L) && n < PYA_MAX_PEPTIDE_LEN) sites[n++] = i;
    std::vector<float> mm(L + 2, 0.f);
    if ((size_t)n_of_mod > n) {
        if (h->allow_n) mm.front() += h->mod_mass;
        else mm.back() += h->mod_mass;
    }
    if (sig_len < 0) sig_len = (int32_t)n;
    for (int32_t j = 0; j < sig_len && j < 64; j++) {
        if (!(sig_bits >> j & 1)) continue;
        size_t pos = (size_t)j < n ? sites[j] : L;
        unsigned char aa = pos < L ? pep[pos] : 0;
        if (aa && h->in_group[aa]) mm[pos + 1] += h->mod_mass;
        else if (pos == 0) mm.front() += h->mod_mass;
        else if (pos + 1 == L) mm.back() += h->mod_mass;
    }
    for (uint64_t a = 0; a < n_aux; a++)
        if (aux_pos[a] < mm.size()) mm[aux_pos[a]] += aux_mass[a];
    size_t s = 0, e = mm.size();
    if (mm.front() == 0.f) s++;
    if (mm.back() == 0.f) e--;
    out.clear();
    for (size_t i = s; i < e; i++) {
        out += i == 0 ? 'n' : (i == L + 1 ? 'c' : (char)pep[i - 1]);
        if (mm[i] > 0.f) {
            char t[16];
            std::snprintf(t, sizeof t, "[%d]", (int)std::round(mm[i]));
            out += t;
        }
    }
}

int pya_format_peptide(const pya_handle *h, const uint8_t *pep, uint64_t L, int32_t n_of_mod,
                       const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux, uint64_t sig_bits,
                       int32_t sig_len, char *buf, uint64_t cap) {
    if (!h || !pep || !buf || cap == 0) return PYA_ERR_ARG;
    if (L > PYA_MAX_PEPTIDE_LEN) return PYA_ERR_LIMIT;
    std::string out;
    format_one(h, pep, L, n_of_mod, aux_pos, aux_mass, n_aux, sig_bits, sig_len, out);
    size_t ncopy = std::min<size_t>(out.size(), cap - 1);
    std::memcpy(buf, out.data(), ncopy);
    buf[ncopy] = 0;
    return (int)out.size();
}

/* the same for many records in one call (threaded above 20 000 records) */
int pya_format_peptides(const pya_handle *h, const pya_batch *b, uint64_t n_rec, const int64_t *rec_psm,
                        const uint64_t *sig_bits, const int32_t *rec_valid, int64_t *str_off, char *buf,
                        uint64_t cap) {
    if (!h || !b || !sig_bits || !str_off) return PYA_ERR_ARG;
    if (b->n_psm && (!b->pep || !b->pep_off || !b->n_of_mod)) return PYA_ERR_ARG;
    const bool has_aux = b->aux_off && b->aux_pos && b->aux_mass;
    std::vector<std::string> strs(n_rec);
    int bad = 0;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t r = lo; r < hi; r++) {
            if (rec_valid && rec_valid[r] <= 0) continue;           /* no localisation: empty string */
            const uint64_t i = rec_psm ? (uint64_t)rec_psm[r] : r;
            if (i >= b->n_psm) {
                bad = 1;
                continue;
            }
            const int64_t p0 = b->pep_off[i], L = b->pep_off[i + 1] - p0;
            if (L < 1 || L > PYA_MAX_PEPTIDE_LEN) continue;         /* set-aside PSM */
            const int64_t a0 = has_aux ? b->aux_off[i] : 0, a1 = has_aux ? b->aux_off[i + 1] : 0;
            format_one(h, b->pep + p0, (uint64_t)L, b->n_of_mod[i], has_aux ? b->aux_pos + a0 : nullptr,
                       has_aux ? b->aux_mass + a0 : nullptr, (uint64_t)(a1 - a0), sig_bits[r], -1, strs[r]);
        }
    };
    unsigned nt = n_rec >= 20000 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (nt == 1) {
        work(0, n_rec);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n_rec * t / nt, n_rec * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
    if (bad) return PYA_ERR_ARG;
    int64_t total = 0;
    for (uint64_t r = 0; r < n_rec; r++) {
        str_off[r] = total;
        total += (int64_t)strs[r].size();
    }
    str_off[n_rec] = total;
    if (cap == 0 || !buf) return PYA_OK;                            /* size query */
    if ((uint64_t)total > cap) return PYA_ERR_ARG;
    for (uint64_t r = 0; r < n_rec; r++) std::memcpy(buf + str_off[r], strs[r].data(), strs[r].size());
    return PYA_OK;
}

} /* extern "C" */

extern "C" {

uint64_t pya_get_workspace_budget(const pya_handle *h) { return h ? (uint64_t)workspace_budget(h) : 0; }

int pya_set_workspace_budget(pya_handle *h, uint64_t bytes) {
    if (!h) return PYA_ERR_ARG;
    if (bytes && bytes < ((uint64_t)16 << 20)) return h->fail(PYA_ERR_ARG, -1, "workspace budget below 16 MiB");
    h->ws_budget = (size_t)bytes;
    return PYA_OK;
}

int pya_last_batch_status(pya_handle *h, int32_t *status, uint64_t n) {
    if (!h || !status) return PYA_ERR_ARG;
    if (h->last_status.empty()) {                       /* nothing was set aside (or the flag was not given) */
        std::memset(status, 0, n * sizeof(int32_t));
        return PYA_OK;
    }
    if (n != h->last_status.size()) return h->fail(PYA_ERR_ARG, -1, "the last batch had %zu PSMs", h->last_status.size());
    std::memcpy(status, h->last_status.data(), n * sizeof(int32_t));
    return PYA_OK;
}

int pya_get_pep_scores_range(pya_handle *h, uint64_t psm_begin, uint64_t psm_end, uint64_t cap, int64_t *rec_off,
                             uint64_t *sig_bits, int32_t *counts, float *scores, float *ws_out,
                             int32_t *nfrag_out) {
    if (!h || !rec_off) return PYA_ERR_ARG;
    pya_plan *p = h->kept;
    if (!p) return h->fail(PYA_ERR_STATE, -1, "no batch retained: call pya_score_batch with PYA_FLAG_KEEP first");
    if (psm_begin > psm_end || psm_end > p->n_psm) return h->fail(PYA_ERR_ARG, -1, "PSM index out of range");
    const int64_t s_begin = p->sig_off[psm_begin], s_end = p->sig_off[psm_end];
    const uint64_t total = (uint64_t)(s_end - s_begin);
    for (uint64_t i = psm_begin; i <= psm_end; i++) rec_off[i - psm_begin] = p->sig_off[i] - s_begin;
    if (total == 0 || cap == 0) return PYA_OK;
    if (cap < total)
        return h->fail(PYA_ERR_ARG, -1, "capacity %llu < %llu records", (unsigned long long)cap,
                       (unsigned long long)total);
    if (!sig_bits || !counts || !scores || !ws_out || !nfrag_out) return h->fail(PYA_ERR_ARG, -1, "NULL output array");
    HIPCHK(h, hipSetDevice(h->device));
    /* one copy per workspace array for the whole range, then the permutation on the host */
    const uint32_t RW = h->rec_words(), NT = h->n_top;
    std::vector<uint32_t> rec((size_t)total * RW), sorted(total);
    std::vector<float> ws(total);
    HIPCHK(h, hipMemcpy(rec.data(), p->d_rec.p + s_begin * RW, rec.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sorted.data(), p->d_sorted.p + s_begin, (size_t)total * 4, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(ws.data(), p->d_ws.p + s_begin, (size_t)total * 4, hipMemcpyDeviceToHost));
    for (uint64_t psm = psm_begin; psm < psm_end; psm++) {
        const uint32_t N = p->n_sig[psm];
        const size_t o = (size_t)(p->sig_off[psm] - s_begin);
        const uint64_t *order = h->order_tab.data() + p->order_off[psm];
        for (uint32_t r = 0; r < N; r++) {
            const uint32_t i = sorted[o + r];
            if (i >= N) return h->fail(PYA_ERR_HIP, (int64_t)psm, "corrupt sort permutation");
            const uint32_t *w = &rec[(o + i) * RW];
            const uint32_t nf = w[RW - 1];
            sig_bits[o + r] = order[i];
            ws_out[o + r] = ws[o + i];
            nfrag_out[o + r] = (int32_t)nf;
            for (uint32_t d = 0; d < NT; d++) {
                const uint32_t c = (w[d >> 1] >> ((d & 1) * 16)) & 0xffffu;
                counts[(o + r) * NT + d] = (int32_t)c;
                /* same table the kernels read (score_table.cpp) */
                scores[(o + r) * NT + d] =
                    nf < h->lut_off.size() ? h->lut[h->lut_off[nf] + (uint32_t)d * (nf + 1) + c] : 0.f;
            }
        }
    }
    return PYA_OK;
}

int pya_get_pep_scores(pya_handle *h, uint64_t psm, uint64_t cap, uint64_t *n_out, uint64_t *sig_bits,
                       int32_t *counts, float *scores, float *ws_out, int32_t *nfrag_out) {
    if (!h || !n_out) return PYA_ERR_ARG;
    int64_t off[2] = {0, 0};
    const int rc = pya_get_pep_scores_range(h, psm, psm + 1, 0, off, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    *n_out = (uint64_t)off[1];
    if (off[1] == 0 || cap == 0) return PYA_OK;
    return pya_get_pep_scores_range(h, psm, psm + 1, cap, off, sig_bits, counts, scores, ws_out, nfrag_out);
}

int pya_calculate_ambiguity(pya_handle *h, uint64_t psm, uint64_t ref_bits, const float *ref_scores,
                            float ref_ws, uint64_t other_bits, const float *other_scores, float other_ws,
                            float *out) {
    if (!h || !ref_scores || !other_scores || !out) return PYA_ERR_ARG;
    pya_plan *p = h->kept;
    if (!p) return h->fail(PYA_ERR_STATE, -1, "no batch retained: call pya_score_batch with PYA_FLAG_KEEP first");
    if (psm >= p->n_psm) return h->fail(PYA_ERR_ARG, -1, "PSM index out of range");
    HIPCHK(h, hipSetDevice(h->device));
    const int64_t L = p->pep_off[psm + 1] - p->pep_off[psm];
    /* (the one-PSM kernel's retained view carries no peak offsets: its spectra are staged in LDS, never big) */
    const int64_t P = p->peak_off.size() > psm + 1 ? p->peak_off[psm + 1] - p->peak_off[psm] : 0;
    const uint32_t NT = h->n_top;                                  /* depth scores per container (Ascore.pyx:196-201) */
    const uint32_t per_type = (uint32_t)std::max<int64_t>(L - 1, 1) * (uint32_t)p->max_charge[psm] * (uint32_t)h->cfg.n_uniq;
    std::vector<float> host_scores(2 * (size_t)NT);
    std::memcpy(host_scores.data(), ref_scores, NT * sizeof(float));
    std::memcpy(host_scores.data() + NT, other_scores, NT * sizeof(float));
    DevBuf<float> d_scores, d_out;
    refresh_shared(p);
    HIPCHK(h, d_scores.upload(host_scores.data(), host_scores.size()));
    HIPCHK(h, d_out.alloc(2));
    int e;
    /* The fast kernel stages the PSM's retained table in LDS sized by the plan's peak_cap, which covers the spectra of
     * up to PYA_FAST_PEAKS peaks only; it takes peptides of up to 64 residues and ten depths.  Everything else -- a long
     * peptide, n_top 11..16, a spectrum binned by the global kernel -- goes through the general kernel's own Ascore
     * code (general_psm.hip), which reads the retained table where it lies. */
    if (L > PYA_FAST_PEPTIDE_LEN || h->all_general() || P > PYA_FAST_PEAKS || per_type > PYA_FAST_FRAGMENTS_PER_TYPE) {
        if (pya_general_lds_bytes((uint32_t)L, per_type) > kMaxLds)
            return h->fail(PYA_ERR_LIMIT, (int64_t)psm, "calculate_ambiguity: %u fragments per ion type exceed the general kernel's room", per_type);
        e = pya_launch_general_ambiguity(&p->dev, (uint32_t)psm, (uint32_t)L, per_type, ref_bits, other_bits, d_scores.p, NT,
                                         ref_ws, other_ws, d_out.p, nullptr);
    } else {
        const uint32_t list_cap = next_pow2(std::max<uint32_t>(1, per_type));
        e = pya_launch_ambiguity(&p->dev, (uint32_t)psm, p->peak_cap, list_cap, ref_bits, other_bits, d_scores.p,
                                 ref_ws, other_ws, d_out.p, nullptr);
    }
    if (e) return h->hip_fail((hipError_t)e, "ambiguity launch");
    float res[2];
    HIPCHK(h, hipMemcpy(res, d_out.p, sizeof res, hipMemcpyDeviceToHost));
    if (res[1] != 0.f) return h->fail(PYA_ERR_LIMIT, (int64_t)psm, "trial count outside the score table");
    *out = res[0];
    return PYA_OK;
}

int pya_debug_wave_ops(pya_handle *h, const int32_t in[64], int32_t out[263]) {
    if (!h || !in || !out) return PYA_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf<int32_t> di, dout;
    HIPCHK(h, di.upload(in, 64));
    HIPCHK(h, dout.alloc(263));
    HIPCHK(h, hipMemset(dout.p, 0, 263 * sizeof(int32_t)));
    int e = pya_launch_debug_wave_ops(di.p, dout.p, nullptr);
    if (e) return h->hip_fail((hipError_t)e, "debug wave ops launch");
    HIPCHK(h, hipMemcpy(out, dout.p, 263 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return PYA_OK;
}

int pya_debug_sort(pya_handle *h, const float *keys, uint32_t n, uint32_t *perm) {
    if (!h || !keys || !perm) return PYA_ERR_ARG;
    if (n == 0) return PYA_OK;
    if (n > PYA_MAX_SIGNATURES) return h->fail(PYA_ERR_LIMIT, -1, "n too large");
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf<float> dk;
    DevBuf<uint32_t> dp;
    HIPCHK(h, dk.upload(keys, n));
    HIPCHK(h, dp.alloc(n));
    int e = pya_launch_debug_sort(dk.p, n, dp.p, nullptr);
    if (e) return h->hip_fail((hipError_t)e, "debug sort launch");
    HIPCHK(h, hipMemcpy(perm, dp.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return PYA_OK;
}

} /* extern "C" */
