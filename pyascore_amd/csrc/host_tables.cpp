/* host_tables.cpp -- environment switches, configuration -> DevConfig, score table, pre-sort order tables of the shapes. */
#include "host_internal.h"

/* The environment reaches the library through FOUR variables, read once per handle (pya_create) and again only when
 * pya_reload_env asks: two sizes a deployment may want to set without touching code, two diagnostics that change no
 * result and no route.  Everything that selects a kernel route or makes a kernel decline work is a DEBUG SWITCH: set
 * per handle through pya_set_debug (include/pyascore_debug.h, test-only), never by a user's shell. */
static void read_env(Knobs &k) {
    auto flag = [](const char *n) { return std::getenv(n) != nullptr; };
    k.host_timing = flag("PYA_HOST_TIMING");
    k.stamps = flag("PYA_STAMPS");
    k.chunk_mb = 0.;
    if (const char *e = std::getenv("PYA_CHUNK_MB")) k.chunk_mb = std::max(1.0, std::atof(e));
    const char *w = std::getenv("PYA_WORKSPACE_MB");
    k.workspace_mb = w ? (int64_t)std::atoll(w) : 0;
}

void read_knobs(Knobs &k) {
    k = Knobs();                 /* every debug switch back to the production default */
    read_env(k);
}

/* One debug switch by name (the names the tests have always used); value == nullptr restores its default.
 * Returns false for a name that is not a switch. */
bool set_knob(Knobs &k, const char *key, const char *value) {
    const Knobs dflt;
    const std::string name(key ? key : "");
    const bool on = value != nullptr;
    auto num = [&](int64_t d) { return on ? (int64_t)std::strtoll(value, nullptr, 0) : d; };
    struct { const char *n; bool Knobs::*m; } flags[] = {
        {"PYA_NO_PLAIN", &Knobs::no_plain}, {"PYA_NO_FUSED", &Knobs::no_fused}, {"PYA_NO_BIG", &Knobs::no_big},
        {"PYA_NO_TINY", &Knobs::no_tiny}, {"PYA_NO_PREFIX", &Knobs::no_prefix}, {"PYA_NO_CHUNKS", &Knobs::no_chunks},
        {"PYA_NO_UPLOAD_THREAD", &Knobs::no_upload_thread}, {"PYA_ONE_PEAK_CLASS", &Knobs::one_peak_class},
        {"PYA_PEAK_CLASSES", &Knobs::peak_classes}, {"PYA_ONE_LDS_CLASS", &Knobs::one_lds_class},
        {"PYA_SORT_ROOM", &Knobs::sort_room}, {"PYA_NO_BIG_INLINE", &Knobs::no_big_inline},
        {"PYA_NO_LOC_HASH", &Knobs::no_loc_hash}, {"PYA_NO_CNT", &Knobs::no_cnt}, {"PYA_NO_NODES", &Knobs::no_nodes},
        {"PYA_HOST_TIMING", &Knobs::host_timing}, {"PYA_STAMPS", &Knobs::stamps},
        {"PYA_SLOW_NULL_STREAM", &Knobs::slow_null_stream}, {"PYA_NO_FORK", &Knobs::no_fork},
    };
    bool known = false;
    for (auto &f : flags)
        if (name == f.n) {
            k.*(f.m) = on;
            known = true;
        }
    if (name == "PYA_DEBUG") k.debug = (uint32_t)num(0), known = true;
    else if (name == "PYA_PLAIN_MIN") k.plain_min = num(dflt.plain_min), known = true;
    else if (name == "PYA_BIG_MIN_N") k.big_min_n = num(dflt.big_min_n), known = true;
    else if (name == "PYA_BIN_SELECT_MIN") k.bin_select_min = num(dflt.bin_select_min), known = true;
    else if (name == "PYA_BIN_SELECT_SCAP") k.bin_select_scap = num(dflt.bin_select_scap), known = true;
    else if (name == "PYA_TINY_MAX") k.tiny_max = num(dflt.tiny_max), known = true;
    else if (name == "PYA_SORT_ROOM_MAX") k.sort_room_max = (uint32_t)num(dflt.sort_room_max), known = true;
    else if (name == "PYA_SB") k.sb = (int)num(-1), known = true;
    else if (name == "PYA_GTP") k.gtp = (int)num(-1), known = true;
    else if (name == "PYA_HASH_PP") k.hash_pp = (int)num(-1), known = true;
    else if (name == "PYA_NODE_CAP") k.node_cap = (int)num(-1), known = true;
    else if (name == "PYA_CHUNK_MB") k.chunk_mb = on ? std::max(1.0, std::atof(value)) : 0., known = true;
    else if (name == "PYA_WORKSPACE_MB") k.workspace_mb = num(0), known = true;
    return known;
}


/* ---------------------------------------------------------------------------------------- */
/* configuration -> DevConfig                                                               */
/* ---------------------------------------------------------------------------------------- */
int build_dev_config(pya_handle *h) {
    DevConfig &c = h->cfg;
    std::memset(&c, 0, sizeof c);
    c.bin_size = h->bin_size;
    c.mod_mass = h->mod_mass;
    c.mz_error = h->mz_error;
    int nt = 0;
    for (char t : h->fragment_types)
        if (is_forward(t)) c.types[nt++] = (uint8_t)t;
    c.n_fwd = nt;
    for (char t : h->fragment_types)
        if (is_backward(t)) c.types[nt++] = (uint8_t)t;
    c.n_types = nt;
    c.first_forward = h->fragment_types.empty() ? 1 : (is_forward(h->fragment_types[0]) ? 1 : 0);
    c.allow_n = h->mod_group.find('n') != std::string::npos;
    c.allow_c = h->mod_group.find('c') != std::string::npos;
    for (int l = 0; l < 26; l++) {
        char up = (char)('A' + l), lowc = (char)('a' + l);
        c.res_mass[l] = std_residue_mass(up);
        c.res_modifiable[l] = h->mod_group.find(up) != std::string::npos;
        (void)lowc;
    }
    /* distinct neutral-loss masses -> classes 1..D */
    std::vector<float> vals;
    auto cls_of = [&](float v) -> int {
        for (size_t i = 0; i < vals.size(); i++)
            if (vals[i] == v) return (int)i + 1;
        vals.push_back(v);
        return (int)vals.size();
    };
    for (int l = 0; l < 26; l++) {
        auto u = h->nl.find((char)('A' + l));
        auto lo = h->nl.find((char)('a' + l));
        /* a zero loss is "no loss" (ModifiedPeptide.cpp:400 tests != 0) */
        if (u != h->nl.end() && u->second != 0.f) c.nl_upper[l] = (uint8_t)cls_of(u->second);
        if (lo != h->nl.end() && lo->second != 0.f) c.nl_lower[l] = (uint8_t)cls_of(lo->second);
    }
    if (vals.size() > PYA_MAX_NL)
        return h->fail(PYA_ERR_LIMIT, -1, "more than %d distinct neutral-loss masses", PYA_MAX_NL);
    c.n_nl = (uint8_t)vals.size();
    /* PowerSetSum(stack, 2): {0} U singles U pair sums, exact-deduplicated (Util.cpp:95-141).
     * Candidate = (value, requirement on the per-class counts). */
    struct Cand {
        float v;
        int a, b;  /* classes (0-based); a == b means the class must occur twice; b < 0: single */
    };
    std::vector<Cand> cands;
    const int D = (int)vals.size();
    for (int a = 0; a < D; a++) cands.push_back({0.f + vals[a], a, -1});
    for (int a = 0; a < D; a++)
        for (int b2 = a; b2 < D; b2++) cands.push_back({(0.f + vals[a]) + vals[b2], a, b2});
    std::vector<float> uniq{0.f};
    std::vector<int> cand_u(cands.size());
    for (size_t i = 0; i < cands.size(); i++) {
        int u = -1;
        for (size_t j = 0; j < uniq.size(); j++)
            if (uniq[j] == cands[i].v) u = (int)j;
        if (u < 0) {
            uniq.push_back(cands[i].v);
            u = (int)uniq.size() - 1;
        }
        cand_u[i] = u;
    }
    if (uniq.size() > PYA_MAX_UNIQ_WIDE || cands.size() > PYA_MAX_NL_CANDS || (D <= PYA_FAST_NL && uniq.size() > PYA_MAX_UNIQ))
        return h->fail(PYA_ERR_LIMIT, -1, "too many neutral-loss sums");
    c.n_uniq = (int32_t)uniq.size();
    c.n_cand = (int32_t)cands.size();
    for (size_t j = 0; j < uniq.size(); j++) c.uniq_w[j] = uniq[j];
    for (size_t i = 0; i < cands.size(); i++) {
        c.cand_a[i] = (uint8_t)cands[i].a;
        c.cand_b[i] = cands[i].b < 0 ? 255 : (uint8_t)cands[i].b;
        c.cand_u[i] = (uint8_t)cand_u[i];
    }
    /* (more than PYA_FAST_NL masses: every PSM goes through the general kernel, which reads the candidates above; the narrow
     * tables stay as the memset at the top left them -- present[st] == 0 is "no state is valid", not even the unmodified ion:
     * a fast kernel routed here by mistake would count nothing instead of reading another configuration's tables) */
    for (size_t j = 0; j < uniq.size() && j < PYA_MAX_UNIQ; j++) c.uniq[j] = uniq[j];
    for (int st = 0; st < 256 && D <= PYA_FAST_NL; st++) {
        int cnt[4];
        bool valid = true;
        for (int a = 0; a < 4; a++) {
            cnt[a] = (st >> (2 * a)) & 3;
            if (cnt[a] == 3 || (a >= D && cnt[a])) valid = false;
        }
        uint16_t m = 1;
        if (valid)
            for (size_t i = 0; i < cands.size(); i++) {
                const Cand &k = cands[i];
                bool ok = k.b < 0 ? cnt[k.a] >= 1 : (k.a == k.b ? cnt[k.a] >= 2 : (cnt[k.a] >= 1 && cnt[k.b] >= 1));
                if (ok) m |= (uint16_t)(1u << cand_u[i]);
            }
        c.present[st] = m;
    }
    /* Ascore.cpp:15-19 */
    const float w[PYA_NTOP] = {0.5f, 0.75f, 1.f, 1.f, 1.f, 1.f, 0.75f, 0.5f, 0.25f, 0.25f};
    double sum = 0.;
    for (float x : w) sum += x;
    float fs = (float)sum;
    for (int i = 0; i < PYA_NTOP; i++) c.weights[i] = w[i] / fs;
    c.n_top = (int32_t)h->n_top;
    return PYA_OK;
}

int sync_config(pya_handle *h) {
    if (!h->cfg_dirty) return PYA_OK;
    int rc = build_dev_config(h);
    if (rc) return rc;
    HIPCHK(h, h->d_cfg.upload(&h->cfg, 1));
    HIPCHK(h, hipDeviceSynchronize());
    h->cfg_dirty = false;
    return PYA_OK;
}

int ensure_lut(pya_handle *h, uint32_t n_max) {
    if (n_max > PYA_MAX_LUT_N)
        return h->fail(PYA_ERR_LIMIT, -1, "a PSM can have up to %u theoretical fragments per site "
                       "assignment; the score table covers %u", n_max, PYA_MAX_LUT_N);
    if (h->lut_uploaded_n > n_max) return PYA_OK;
    uint32_t target = std::max<uint32_t>(n_max, 128);
    pya_score_table_extend(h->mz_error, h->n_top, target, h->lut, h->lut_off);
    for (size_t n = 0; n < h->lut_off.size(); n++)             /* the kernels compute row offsets */
        if (h->lut_off[n] != h->n_top * (uint32_t)n * ((uint32_t)n + 1u) / 2u)
            return h->fail(PYA_ERR_STATE, -1, "score table rows are not dense");
    HIPCHK(h, h->d_lut.upload(h->lut.data(), h->lut.size()));
    HIPCHK(h, h->d_lut_off.upload(h->lut_off.data(), h->lut_off.size()));
    HIPCHK(h, hipDeviceSynchronize());
    h->lut_uploaded_n = target + 1;
    return PYA_OK;
}

/* Pre-sort order of the signatures of a shape: keys (N-term site = MSB) are inserted into the
 * reference's hash map in the first fragment type's traversal order and read back in the
 * container's iteration order (cpp/Ascore.cpp:91-120, cpp/ModifiedPeptide.cpp:410-476). */
uint32_t shape_offset(pya_handle *h, uint32_t n, uint32_t k) {
    uint32_t key = n << 8 | k;
    auto it = h->shape_off.find(key);
    if (it != h->shape_off.end()) return it->second;
    uint32_t off = (uint32_t)h->order_tab.size();
    const bool fwd = h->cfg.first_forward;
    std::unordered_map<long, uint64_t> order;
    if (k <= n) {
        std::vector<uint32_t> c(k);
        for (uint32_t i = 0; i < k; i++) c[i] = i;
        for (;;) {
            uint64_t bits = 0;
            for (uint32_t t : c) bits |= 1ull << (fwd ? t : n - 1 - t);
            long lk = 0;
            for (uint32_t j = 0; j < n; j++) lk = (lk << 1) | (long)(bits >> j & 1);
            order.emplace(lk, bits);
            int j = (int)k - 1;
            while (j >= 0 && c[j] == n - k + (uint32_t)j) j--;
            if (j < 0) break;
            c[j]++;
            for (uint32_t t = j + 1; t < k; t++) c[t] = c[t - 1] + 1;
        }
    }
    for (auto &kv : order) h->order_tab.push_back(kv.second);
    /* inverse: colexicographic rank of a signature (sum over its set bits of C(position, ordinal)) ->
     * where the signature sits in the pre-sort order; localize enumerates single-move competitors with it */
    h->inv_tab.resize(h->order_tab.size(), 0u);
    for (size_t i = off; i < h->order_tab.size(); i++) {
        uint64_t m = h->order_tab[i];
        uint64_t rank = 0;
        for (uint32_t t = 1; m; t++) {
            const uint32_t pos = (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            rank += binom(pos, t);
        }
        h->inv_tab[off + rank] = (uint32_t)(i - off);
    }
    /* Shared-node table of the shape (score_core.hip.h: score_nodes_dir), behind its order entries, for shapes of
     * at most 64 signatures: per direction and level j (sites passed: the lowest j in direction 0, the highest j
     * in direction 1) the signatures that are the lowest of their group -- same pattern over those sites -- and
     * every signature's group rank as a byte. */
    const size_t N = h->order_tab.size() - off;
    if (N >= 1 && N <= 64) {
        const size_t W8 = (N + 7) / 8;
        std::vector<uint64_t> own(2 * (n + 1), 0ull), grp(2 * (n + 1) * W8, 0ull);
        for (uint32_t dir = 0; dir < 2; dir++)
            for (uint32_t j = 0; j <= n; j++) {
                std::vector<uint64_t> seen;
                uint8_t *row = (uint8_t *)(grp.data() + (size_t)(dir * (n + 1) + j) * W8);
                for (size_t sidx = 0; sidx < N; sidx++) {
                    const uint64_t bits = h->order_tab[off + sidx];
                    const uint64_t pat = j == 0 ? 0ull : (dir == 0 ? (bits & ((j >= 64 ? 0ull : (1ull << j)) - 1ull)) : (bits >> (n - j)));
                    size_t g = 0;
                    while (g < seen.size() && seen[g] != pat) g++;
                    if (g == seen.size()) {
                        seen.push_back(pat);
                        own[dir * (n + 1) + j] |= 1ull << sidx;
                    }
                    row[sidx] = (uint8_t)g;
                }
            }
        uint32_t cols[2] = {0, 0};
        for (uint32_t dir = 0; dir < 2; dir++)
            for (uint32_t j = 0; j <= n; j++) cols[dir] += (uint32_t)__builtin_popcountll(own[dir * (n + 1) + j]);
        h->shape_cols[key] = std::max(cols[0], cols[1]);
        h->order_tab.insert(h->order_tab.end(), own.begin(), own.end());
        h->order_tab.insert(h->order_tab.end(), grp.begin(), grp.end());
        h->inv_tab.resize(h->order_tab.size(), 0u);
    }
    h->shape_off[key] = off;
    return off;
}

uint32_t next_pow2(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

/* tables owned by the handle may have been re-uploaded (grown) since the plan was made */
