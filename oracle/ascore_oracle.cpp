/*
 * ascore_oracle.cpp -- TEST INFRASTRUCTURE ONLY (CPU restatement; never shipped, never the
 * thing measured except as bench.py's "cpu_baseline", never imported by pyascore_amd/).
 *
 * A from-the-spec restatement of pyAscore's PyAscore.score hot path:
 * the same arithmetic, in the same float/double order, but structured the way the MI355X
 * kernels are structured -- independent per-signature fragment walks against a retained-peak
 * table, instead of the reference's hash-map match cache and prefix-sharing iterator.
 *
 * PARITY PINNING: checked against (1) oracle/_ref (the reference's own C++ core compiled in
 * this container, tests/test_oracle_vs_ref.py), (2) the committed golden vectors in
 * tests/golden/ that were generated from oracle/_ref, and (3) the literal known answers of the
 * reference's unit tests (tests/test_oracle_known_answers.py).
 *
 * Reference citations are into /root/reference/pyascore/ptm_scoring/cpp/.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "oracle_abi.h"

namespace {

/* ------------------------------------------------------------------------------------------
 * Constants.  Types.h:7-30 (residue table is float32), ModifiedPeptide.cpp:572-587.
 * ---------------------------------------------------------------------------------------- */
float residue_mass(char c, bool *ok) {
    *ok = true;
    switch (c) {
        case 'G': return 57.02146f;   case 'A': return 71.03711f;   case 'S': return 87.03203f;
        case 'P': return 97.05276f;   case 'V': return 99.06841f;   case 'T': return 101.04768f;
        case 'C': return 103.00919f;  case 'L': return 113.08406f;  case 'I': return 113.08406f;
        case 'N': return 114.04293f;  case 'D': return 115.02694f;  case 'Q': return 128.05858f;
        case 'K': return 128.09496f;  case 'E': return 129.04259f;  case 'M': return 131.04049f;
        case 'H': return 137.05891f;  case 'F': return 147.06841f;  case 'U': return 150.95364f;
        case 'R': return 156.10111f;  case 'Y': return 163.06333f;  case 'W': return 186.07931f;
        case 'O': return 237.14773f;
    }
    *ok = false;
    return 0.f;
}
const double H2O = 18.010565, NH3 = 17.026549, NH2 = 16.018724, PROTON = 1.007825;

bool forward_type(char t) { return t == 'b' || t == 'c'; }
bool known_type(char t) { return t == 'b' || t == 'c' || t == 'y' || t == 'z' || t == 'Z'; }

/* ------------------------------------------------------------------------------------------
 * Binomial chain, float32 in the reference's exact operation order.  Util.cpp:16-83.
 * ---------------------------------------------------------------------------------------- */
float log_sum(float a, float b) {                       /* Util.cpp:16-26 */
    if (std::isinf(a)) return b;
    if (std::isinf(b)) return a;
    float m = std::max(a, b);
    float s = std::exp(a - m) + std::exp(b - m);        /* expf, float add */
    return m + std::log(s);                             /* logf */
}

float log_bin_coef(uint64_t k, uint64_t n) {            /* Util.cpp:28-41 */
    float coef = 0.f;
    k = std::min(n - k, k);
    for (uint64_t m = n - k + 1; m <= n; m++) coef = (float)((double)coef + std::log((double)m));
    for (uint64_t m = 2; m <= k; m++) coef = (float)((double)coef - std::log((double)m));
    return coef;
}

struct Binom {                                           /* Util.cpp:47-59 */
    float log_p, log_q;
    explicit Binom(float p) {
        log_p = std::log(p);                             /* logf(float) */
        log_q = (float)std::log(1. - (double)p);         /* double log, narrowed */
    }
    float log_pmf(uint64_t k, uint64_t n) const {
        float a = log_bin_coef(k, n);
        float b = (float)k * log_p;
        float c = (float)(n - k) * log_q;
        return (a + b) + c;
    }
    /* tail[k] = log P(X >= k), k = 0..n ; tail[0] := 0 as in Util.cpp:65 */
    std::vector<float> tail(uint64_t n) const {
        std::vector<float> t(n + 2);
        t[n + 1] = -INFINITY;
        for (uint64_t j = n; j >= 1; j--) t[j] = log_sum(t[j + 1], log_pmf(j, n));
        t[0] = 0.f;
        return t;
    }
};

float log10_of(float log_pvalue) {                       /* Util.cpp:81-83 */
    return (float)(std::log10(std::exp(1.0)) * (double)log_pvalue);
}

/* PowerSetSum(target, depth): sorted, exactly-deduplicated sums of <= depth elements.
 * Util.cpp:95-141.  Only depth <= 2 is ever used by the hot path (ModifiedPeptide.cpp:404). */
void subset_sums(const std::vector<float> &t, size_t depth, size_t start, float base, size_t d,
                 std::vector<float> &out) {
    for (size_t i = start; i < t.size(); i++) {
        float s = base + t[i];
        out.push_back(s);
        if (d + 1 < depth && i + 1 < t.size()) subset_sums(t, depth, i + 1, s, d + 1, out);
    }
}
std::vector<float> power_set_sums(const std::vector<float> &t, size_t depth) {
    depth = std::min(depth, t.size());
    std::vector<float> out{0.f};
    if (depth > 0) subset_sums(t, depth, 0, 0.f, 0, out);
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

/* ------------------------------------------------------------------------------------------
 * Retained-peak table.  Spectra.cpp:43-68 (binning) and :24-41 (top-n per bin).
 * ---------------------------------------------------------------------------------------- */
struct Peak {
    double mz, intensity;
};
struct Spectrum {
    float bin_size;
    size_t n_top;
    float min_mz = 0.f, max_mz = 0.f;
    size_t n_bins = 0;
    std::vector<std::vector<Peak>> bins;

    bool consume(const double *mz, const double *inten, size_t n) {
        if (n == 0) return false;
        double lo = *std::min_element(mz, mz + n), hi = *std::max_element(mz, mz + n);
        min_mz = (float)(std::floor(lo / 100.) * 100.);  /* the 100 is hard-coded, :46-47 */
        max_mz = (float)(std::ceil(hi / 100.) * 100.);
        n_bins = (size_t)std::ceil((max_mz - min_mz) / bin_size);   /* float arithmetic, :48 */
        if (n_bins == 0) return false;
        bins.assign(n_bins, {});
        for (size_t i = 0; i < n; i++) {
            size_t b = (size_t)std::floor((mz[i] - (double)min_mz) / (double)bin_size);
            bins[std::min(b, n_bins - 1)].push_back({mz[i], inten[i]});
        }
        auto brighter = [](const Peak &a, const Peak &b) { return a.intensity > b.intensity; };
        for (auto &bin : bins) {
            if (n_top < bin.size()) {
                std::nth_element(bin.begin(), bin.begin() + n_top - 1, bin.end(), brighter);
                bin.resize(n_top);
            }
            std::sort(bin.begin(), bin.end(), brighter);
        }
        return true;
    }
};

/* Net semantics of the match cache (consumePeak / hasMatch / getMatch,
 * ModifiedPeptide.cpp:126-150 driven by Ascore.pyx:142-150):
 *   match(f) = min rank over retained peaks p=(float)mz with  f32(f-err) < p < f32(f+err)
 *              and f >= p - 0.5 (the lower_bound start at :128; vacuous for err <= 0.5 up to
 *              one rounding), else "none".                                                   */
struct MatchTable {
    struct Entry { float mz; int rank; };
    std::vector<Entry> peaks;                            /* sorted by mz */
    float err;
    void build(const Spectrum &s, float mz_error) {
        err = mz_error;
        peaks.clear();
        for (const auto &bin : s.bins)
            for (size_t r = 0; r < bin.size(); r++) peaks.push_back({(float)bin[r].mz, (int)r});
        std::stable_sort(peaks.begin(), peaks.end(),
                         [](const Entry &a, const Entry &b) { return a.mz < b.mz; });
    }
    int rank_of(float f) const {                         /* -1 = no match */
        float lo = f - err, hi = f + err;
        auto it = std::upper_bound(peaks.begin(), peaks.end(), lo,
                                   [](float v, const Entry &e) { return v < e.mz; });
        int best = -1;
        for (; it != peaks.end() && it->mz < hi; ++it) {
            if ((double)f < (double)it->mz - .5) continue;
            if (best < 0 || it->rank < best) best = it->rank;
        }
        return best;
    }
};

/* ------------------------------------------------------------------------------------------
 * Modified peptide.  ModifiedPeptide.cpp:24-79 (residues, NL, aux mods), :570-591 (m/z).
 * ---------------------------------------------------------------------------------------- */
struct Peptide {
    std::string seq;
    size_t n_of_mod = 0, max_charge = 1;
    std::vector<uint32_t> aux_pos;
    std::vector<float> aux_mass;
    std::vector<float> mass0, mass1, nl0, nl1;           /* per residue: unmodified / modified */
    std::vector<char> modifiable, nl1_valid;
    std::vector<size_t> sites;                           /* residue index of each modifiable, N->C */

    bool build(const std::string &mod_group, float mod_mass, const std::map<char, float> &nl,
               const std::string &peptide, size_t k, size_t zmax, const uint32_t *ap,
               const float *am, size_t na) {
        seq = peptide;
        n_of_mod = k;
        max_charge = zmax;
        aux_pos.assign(ap, ap + na);
        aux_mass.assign(am, am + na);
        size_t L = seq.size();
        mass0.assign(L, 0.f); mass1.assign(L, 0.f); nl0.assign(L, 0.f); nl1.assign(L, 0.f);
        modifiable.assign(L, 0); nl1_valid.assign(L, 1);
        sites.clear();
        bool has_n = mod_group.find('n') != std::string::npos;
        bool has_c = mod_group.find('c') != std::string::npos;
        for (size_t i = 0; i < L; i++) {
            bool ok;
            mass0[i] = residue_mass(seq[i], &ok);
            if (!ok) return false;
            auto u = nl.find(seq[i]);
            if (u != nl.end()) nl0[i] = u->second;
            bool m = mod_group.find(seq[i]) != std::string::npos || (has_n && i == 0) ||
                     (has_c && i + 1 == L);
            if (m) {
                modifiable[i] = 1;
                mass1[i] = mass0[i] + mod_mass;
                auto l = nl.find((char)std::tolower(seq[i]));
                if (l != nl.end()) nl1[i] = l->second;
            }
        }
        for (size_t a = 0; a < na; a++) {                /* ModifiedPeptide.cpp:59-79 */
            size_t i = ap[a] > 0 ? ap[a] - 1 : 0;
            if (i >= L) return false;
            mass0[i] += am[a];
            if (modifiable[i]) mass1[i] += am[a];
            auto l = nl.find((char)std::tolower(seq[i]));
            if (l != nl.end()) {
                nl0[i] = l->second;
                nl1_valid[i] = 0;  /* reference resizes the NL vector to 1: state 1 is out of bounds */
            }
        }
        for (size_t i = 0; i < L; i++)
            if (modifiable[i]) sites.push_back(i);
        return true;
    }
    size_t n_sites() const { return sites.size(); }
};

struct Fragment {
    float mz;
    int size;      /* residues in the fragment */
    int loss_pos;  /* 0 = no neutral loss */
};

float fragment_mz(float running, float loss, char type, size_t charge) {
    double m = (double)(running - loss);                 /* float subtract, then widen (:572) */
    if (type == 'y') m += H2O;
    else if (type == 'z') { m += H2O; m -= NH3; }
    else if (type == 'Z') { m += H2O; m -= NH2; }
    else if (type == 'c') m += NH3;
    if (charge > 0) m = (m + (double)charge * PROTON) / (double)charge;
    return (float)m;
}

/* All fragments of one signature (modified[i] per residue) for (type, charge), in the
 * reference's walk order: sizes 1..L-1, per size every NL variant ascending.
 * ModifiedPeptide.cpp:379-408 (running float sum, NL stack), :500-524 (no full-length). */
void walk(const Peptide &p, const std::vector<char> &modified, char type, size_t charge,
          std::vector<Fragment> &out) {
    size_t L = p.seq.size();
    bool fwd = forward_type(type);
    float running = 0.f;
    std::vector<float> stack;
    std::vector<float> sums{0.f};
    for (size_t step = 0; step + 1 < L; step++) {
        size_t i = fwd ? step : L - 1 - step;
        bool m = modified[i];
        float r = m ? p.mass1[i] : p.mass0[i];
        running = step == 0 ? r : r + running;           /* new = residue + running_sum.back() */
        float nl = m ? p.nl1[i] : p.nl0[i];
        if (nl != 0.f) {
            stack.push_back(nl);
            sums = power_set_sums(stack, 2);
        }
        for (size_t v = 0; v < sums.size(); v++)
            out.push_back({fragment_mz(running, sums[v], type, charge), (int)step + 1, (int)v});
    }
}

/* Lexicographic k-combinations of n sites, in traversal-direction site indices.
 * ModifiedPeptide.cpp:410-476 (first k set; right-most movable moves, tail packs after it). */
std::vector<std::vector<int>> combinations(size_t n, size_t k) {
    std::vector<std::vector<int>> out;
    if (k > n) return out;
    std::vector<int> c(k);
    for (size_t i = 0; i < k; i++) c[i] = (int)i;
    for (;;) {
        out.push_back(c);
        int j = (int)k - 1;
        while (j >= 0 && c[j] == (int)(n - k) + j) j--;
        if (j < 0) break;
        c[j]++;
        for (size_t t = j + 1; t < k; t++) c[t] = c[t - 1] + 1;
    }
    return out;
}

/* signature bits (N->C, site j = bit j) of combination c expressed in traversal order */
uint64_t combo_bits(const std::vector<int> &c, size_t n, bool fwd) {
    uint64_t b = 0;
    for (int t : c) b |= 1ull << (fwd ? (size_t)t : n - 1 - (size_t)t);
    return b;
}

std::vector<char> residue_flags(const Peptide &p, uint64_t bits) {
    std::vector<char> f(p.seq.size(), 0);
    for (size_t j = 0; j < p.sites.size(); j++)
        if (bits >> j & 1) f[p.sites[j]] = 1;
    return f;
}

/* ModifiedPeptide.cpp:199-253 */
std::string format_peptide(const Peptide &p, const std::string &mod_group, float mod_mass,
                           std::vector<int> sig) {
    size_t n = p.n_sites(), L = p.seq.size();
    if (sig.empty()) sig.assign(std::min(p.n_of_mod, n), 1);
    std::vector<float> mm(L + 2, 0.f);
    if (p.n_of_mod > n) {
        if (mod_group.find('n') != std::string::npos) mm.front() += mod_mass;
        else mm.back() += mod_mass;
    }
    for (size_t j = 0; j < sig.size(); j++) {
        if (sig[j] != 1) continue;
        size_t pos = j < n ? p.sites[j] : L;             /* getPosOfNthModifiable */
        char aa = pos < L ? p.seq[pos] : 0;
        if (mod_group.find(aa) != std::string::npos) mm[pos + 1] += mod_mass;
        else if (pos == 0) mm.front() += mod_mass;
        else if (pos + 1 == L) mm.back() += mod_mass;
    }
    for (size_t a = 0; a < p.aux_pos.size(); a++) mm[p.aux_pos[a]] += p.aux_mass[a];
    size_t s = 0, e = mm.size();
    if (mm.front() == 0.f) s++;
    if (mm.back() == 0.f) e--;
    std::string full = "n" + p.seq + "c", out;
    for (size_t i = s; i < e; i++) {
        out += full[i];
        if (mm[i] > 0.f) {
            char buf[16];
            std::snprintf(buf, sizeof buf, "[%d]", (int)std::round(mm[i]));
            out += buf;
        }
    }
    return out;
}

/* ------------------------------------------------------------------------------------------
 * Scorer state.  Ascore.cpp.
 * ---------------------------------------------------------------------------------------- */
struct PepScore {
    uint64_t bits;                    /* site j modified = bit j, N-term site = bit 0 */
    std::vector<int64_t> counts;      /* cumulative over rank */
    std::vector<float> scores;
    float weighted = -1.f;
    int64_t nfrag = 0;
};
struct SiteResult {
    std::vector<uint32_t> positions;  /* 1-based peptide positions of tied best competitors */
    std::vector<float> pep_scores, ascores;
};

} /* namespace */

struct orc_handle {
    float bin_size;
    size_t n_top;
    std::string mod_group, ftypes;
    float mod_mass, mz_error;
    std::map<char, float> nl;
    Spectrum spec;
    Peptide pep;
    MatchTable table;
    std::vector<float> weights;
    std::vector<Binom> dists;
    std::map<std::pair<size_t, size_t>, std::vector<float>> tails;   /* (depth, n) -> tail */
    std::vector<PepScore> scores;
    std::vector<SiteResult> site_results;

    orc_handle(float bs, size_t nt, const char *mg, float mm, float err, const char *ft)
        : bin_size(bs), n_top(nt), mod_group(mg), ftypes(ft), mod_mass(mm), mz_error(err) {
        spec.bin_size = bs;
        spec.n_top = nt;
        /* Ascore.cpp:15-19: float weights, double sum, float divide */
        weights = {0.5f, 0.75f, 1.f, 1.f, 1.f, 1.f, 0.75f, 0.5f, 0.25f, 0.25f};
        double s = 0.;
        for (float w : weights) s += w;
        float fs = (float)s;
        for (float &w : weights) w /= fs;
        /* Ascore.cpp:33: (2 * err) float, * depth float, / 100. double, narrowed to float */
        for (size_t d = 1; d <= n_top; d++) {
            float t = (2 * mz_error) * (float)d;
            dists.emplace_back((float)((double)t / 100.));
        }
    }

    float depth_score(size_t depth_idx, int64_t k, int64_t n) {      /* Ascore.cpp:127-133 */
        if (k > n) throw 10;
        auto key = std::make_pair(depth_idx, (size_t)n);
        auto it = tails.find(key);
        if (it == tails.end()) it = tails.emplace(key, dists[depth_idx].tail((uint64_t)n)).first;
        float l10 = log10_of(it->second[(size_t)k]);
        return std::abs(-10 * l10);
    }

    std::vector<int> bits_to_sig(uint64_t bits) const {
        std::vector<int> s(pep.n_sites());
        for (size_t j = 0; j < s.size(); j++) s[j] = (int)(bits >> j & 1);
        return s;
    }

    /* Ascore.cpp:53-121 restated: every signature scored independently. */
    void accumulate() {
        scores.clear();
        size_t n = pep.n_sites();
        if (ftypes.empty() || pep.max_charge == 0) return;
        for (char t : ftypes)
            if (!known_type(t)) throw 30;
        /* pre-sort order = iteration order of the reference's unordered_map<long,...> after
         * inserting keys (N-term site = MSB) in the first fragment type's traversal order */
        bool fwd0 = forward_type(ftypes[0]);
        auto combos = combinations(n, pep.n_of_mod);
        std::unordered_map<long, uint64_t> order;
        for (auto &c : combos) {
            uint64_t bits = combo_bits(c, n, fwd0);
            long key = 0;
            for (size_t j = 0; j < n; j++) key = (key << 1) | (long)(bits >> j & 1);
            order.emplace(key, bits);
        }
        std::vector<Fragment> frags;
        for (auto &kv : order) {
            PepScore ps;
            ps.bits = kv.second;
            ps.counts.assign(n_top, 0);
            std::vector<char> flags = residue_flags(pep, ps.bits);
            for (char t : ftypes)
                for (size_t z = 1; z <= pep.max_charge; z++) {
                    frags.clear();
                    walk(pep, flags, t, z, frags);
                    for (const Fragment &f : frags) {
                        int r = table.rank_of(f.mz);
                        if (r >= 0) ps.counts[r]++;
                        ps.nfrag++;
                    }
                }
            for (size_t d = 1; d < n_top; d++) ps.counts[d] += ps.counts[d - 1];
            scores.push_back(std::move(ps));
        }
    }

    void full_scores() {                                 /* Ascore.cpp:123-139 */
        for (PepScore &ps : scores) {
            ps.scores.resize(n_top);
            for (size_t d = 0; d < n_top; d++) ps.scores[d] = depth_score(d, ps.counts[d], ps.nfrag);
            double acc = 0.;
            for (size_t i = 0; i < weights.size(); i++) acc = acc + (double)(weights[i] * ps.scores[i]);
            ps.weighted = (float)acc;
        }
    }

    /* ModifiedPeptide.cpp:259-320 */
    void site_determining(uint64_t bits1, uint64_t bits2, char type, size_t zmax,
                          std::vector<float> out[2]) {
        std::vector<Fragment> f1, f2;
        std::vector<char> a = residue_flags(pep, bits1), b = residue_flags(pep, bits2);
        for (size_t z = 1; z <= zmax; z++) {
            walk(pep, a, type, z, f1);
            walk(pep, b, type, z, f2);
        }
        std::vector<float> x, y;
        for (auto &f : f1) x.push_back(f.mz);
        for (auto &f : f2) y.push_back(f.mz);
        std::sort(x.begin(), x.end());
        std::sort(y.begin(), y.end());
        size_t i = 0, j = 0;
        out[0].clear();
        out[1].clear();
        while (i < x.size() || j < y.size()) {
            if (j == y.size()) out[0].push_back(x[i++]);
            else if (i == x.size()) out[1].push_back(y[j++]);
            else if (std::abs(x[i] - y[j]) < mz_error) { i++; j++; }
            else if (x[i] < y[j]) out[0].push_back(x[i++]);
            else out[1].push_back(y[j++]);
        }
    }

    /* Ascore.cpp:157-210 */
    float ambiguity(uint64_t ref_bits, const float *ref_scores, float ref_ws, uint64_t oth_bits,
                    const float *oth_scores, float oth_ws, size_t n_scores) {
        if (std::abs(ref_ws - oth_ws) < 1e-6) return 0.f;
        float best = 0.f;
        size_t depth = 0;
        for (size_t d = 0; d < n_scores; d++) {
            float diff = ref_scores[d] - oth_scores[d];
            if (diff > best) { best = diff; depth = d; }
        }
        int64_t cnt[2] = {0, 0}, trials[2] = {0, 0};
        std::vector<float> ions[2];
        for (char t : ftypes) {
            site_determining(ref_bits, oth_bits, t, pep.max_charge, ions);
            for (int s = 0; s < 2; s++) {
                trials[s] += (int64_t)ions[s].size();
                for (float mz : ions[s]) {
                    int r = table.rank_of(mz);
                    if (r >= 0 && (size_t)r <= depth) cnt[s]++;
                }
            }
        }
        float s0 = depth_score(depth, cnt[0], trials[0]);
        float s1 = depth_score(depth, cnt[1], trials[1]);
        return s0 - s1;
    }

    void run() {                                         /* Ascore.cpp:256-271 */
        site_results.clear();
        table.build(spec, mz_error);
        accumulate();
        full_scores();
        size_t n = pep.n_sites(), k = pep.n_of_mod;
        if (k >= n) {                                    /* Ascore.cpp:38-51 */
            for (size_t i = 0; i < k; i++) {
                SiteResult r;
                r.ascores.push_back(std::numeric_limits<float>::infinity());
                site_results.push_back(r);
            }
            return;
        }
        std::sort(scores.begin(), scores.end(),
                  [](const PepScore &a, const PepScore &b) { return a.weighted > b.weighted; });
        if (scores.empty()) throw 60;                    /* reference: UB (front() of empty) */
        const PepScore &best = scores.front();
        site_results.assign(k, {});
        for (const PepScore &c : scores) {               /* Ascore.cpp:212-254 */
            if (k - (size_t)__builtin_popcountll(best.bits & c.bits) != 1) continue;
            size_t same = 0, which = 0, dest = 0;
            for (size_t j = 0; j < n; j++) {
                bool b = best.bits >> j & 1, o = c.bits >> j & 1;
                if (b && o) same++;
                else if (b && !o) which = same;
                else if (!b && o) dest = j;
            }
            SiteResult &r = site_results[which];
            if (r.ascores.empty() || c.weighted == r.pep_scores.back()) {
                r.positions.push_back((uint32_t)pep.sites[dest] + 1);
                r.pep_scores.push_back(c.weighted);
                r.ascores.push_back(ambiguity(best.bits, best.scores.data(), best.weighted, c.bits,
                                              c.scores.data(), c.weighted, best.scores.size()));
            }
        }
    }
};

static uint64_t copy_str(const std::string &s, char *buf, uint64_t cap) {
    if (buf && cap) {
        uint64_t n = std::min<uint64_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}
static uint64_t sig_to_bits(const int32_t *sig, uint64_t n) {
    uint64_t b = 0;
    for (uint64_t j = 0; j < n && j < 64; j++)
        if (sig[j]) b |= 1ull << j;
    return b;
}

extern "C" {

const char *orc_impl_name(void) { return "oracle"; }

orc_handle *orc_create(float bin_size, uint64_t n_top, const char *mod_group, float mod_mass,
                       float mz_error, const char *fragment_types) {
    return new orc_handle(bin_size, n_top, mod_group, mod_mass, mz_error, fragment_types);
}
void orc_destroy(orc_handle *h) { delete h; }
void orc_add_neutral_loss(orc_handle *h, const char *group, float mass) {
    for (const char *c = group; *c; c++) h->nl[*c] = mass;      /* ModifiedPeptide.cpp:99-103 */
}

int orc_consume_spectra(orc_handle *h, const double *mz, const double *inten, uint64_t n) {
    return h->spec.consume(mz, inten, n) ? 0 : -1;
}
int orc_consume_peptide(orc_handle *h, const char *peptide, uint64_t n_of_mod, uint64_t zmax,
                        const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux) {
    if (!aux_pos || !aux_mass) n_aux = 0;
    return h->pep.build(h->mod_group, h->mod_mass, h->nl, peptide, n_of_mod, zmax, aux_pos,
                        aux_mass, n_aux) ? 0 : -1;
}

int orc_score(orc_handle *h, const double *mz, const double *inten, uint64_t n_peaks,
              const char *peptide, uint64_t n_of_mod, uint64_t zmax, const uint32_t *aux_pos,
              const float *aux_mass, uint64_t n_aux) {
    try {
        if (orc_consume_spectra(h, mz, inten, n_peaks)) return -2;
        if (orc_consume_peptide(h, peptide, n_of_mod, zmax, aux_pos, aux_mass, n_aux)) return -3;
        h->run();
    } catch (...) {
        return -1;
    }
    return 0;
}

uint64_t orc_n_pep_scores(orc_handle *h) { return h->scores.size(); }
uint64_t orc_sig_len(orc_handle *h) { return h->pep.n_sites(); }
uint64_t orc_n_top(orc_handle *h) { return h->n_top; }

void orc_get_pep_scores(orc_handle *h, int32_t *sig, int32_t *counts, float *scores, float *ws,
                        int64_t *nfrag) {
    size_t sl = h->pep.n_sites(), nt = h->n_top;
    for (size_t i = 0; i < h->scores.size(); i++) {
        const PepScore &p = h->scores[i];
        for (size_t j = 0; j < sl; j++) sig[i * sl + j] = (int32_t)(p.bits >> j & 1);
        for (size_t d = 0; d < nt; d++) {
            counts[i * nt + d] = (int32_t)p.counts[d];
            scores[i * nt + d] = p.scores[d];
        }
        ws[i] = p.weighted;
        nfrag[i] = p.nfrag;
    }
}

float orc_best_score(orc_handle *h) { return h->scores.empty() ? -1.f : h->scores.front().weighted; }
uint64_t orc_sequence(orc_handle *h, uint64_t idx, char *buf, uint64_t cap) {
    if (idx >= h->scores.size()) return copy_str("", buf, cap);
    return copy_str(format_peptide(h->pep, h->mod_group, h->mod_mass,
                                   h->bits_to_sig(h->scores[idx].bits)), buf, cap);
}
uint64_t orc_best_sequence(orc_handle *h, char *buf, uint64_t cap) {
    return orc_sequence(h, 0, buf, cap);
}
uint64_t orc_n_ascores(orc_handle *h) { return h->site_results.size(); }
void orc_get_ascores(orc_handle *h, float *out) {
    for (size_t i = 0; i < h->site_results.size(); i++) {
        const auto &a = h->site_results[i].ascores;
        out[i] = *std::min_element(a.begin(), a.end());  /* Ascore.cpp:305-313 */
    }
}
uint64_t orc_alt_sites(orc_handle *h, uint64_t site, uint32_t *buf, uint64_t cap) {
    if (site >= h->site_results.size()) return 0;
    std::vector<uint32_t> v = h->site_results[site].positions;
    std::sort(v.begin(), v.end());                       /* Ascore.cpp:315-319 */
    for (size_t i = 0; i < v.size() && i < cap; i++) buf[i] = v[i];
    return v.size();
}

float orc_calculate_ambiguity(orc_handle *h, const int32_t *sig_ref, const float *scores_ref,
                              float ws_ref, const int32_t *sig_other, const float *scores_other,
                              float ws_other, uint64_t sig_len, uint64_t n_scores) {
    try {
        return h->ambiguity(sig_to_bits(sig_ref, sig_len), scores_ref, ws_ref,
                            sig_to_bits(sig_other, sig_len), scores_other, ws_other, n_scores);
    } catch (...) {
        return std::numeric_limits<float>::quiet_NaN();
    }
}

int64_t orc_score_batch(orc_handle *h, uint64_t n_psm, const double *mz, const double *inten,
                        const int64_t *peak_off, const char *pep, const int64_t *pep_off,
                        const int32_t *n_of_mod, const int32_t *max_charge,
                        const uint32_t *aux_pos, const float *aux_mass, const int64_t *aux_off,
                        uint64_t max_k, float *best_score, uint64_t *best_sig, int32_t *n_sig,
                        float *ascores, uint64_t *alt_mask) {
    for (uint64_t i = 0; i < n_psm; i++) {
        std::string p(pep + pep_off[i], pep + pep_off[i + 1]);
        uint64_t na = aux_off ? (uint64_t)(aux_off[i + 1] - aux_off[i]) : 0;
        const uint32_t *ap = na ? aux_pos + aux_off[i] : nullptr;
        const float *am = na ? aux_mass + aux_off[i] : nullptr;
        int rc = orc_score(h, mz + peak_off[i], inten + peak_off[i],
                           (uint64_t)(peak_off[i + 1] - peak_off[i]), p.c_str(),
                           (uint64_t)n_of_mod[i], (uint64_t)max_charge[i], ap, am, na);
        if (rc) return -(int64_t)(i + 1);
        best_score[i] = orc_best_score(h);
        n_sig[i] = (int32_t)h->scores.size();
        best_sig[i] = h->scores.empty() ? 0 : h->scores.front().bits;
        for (uint64_t j = 0; j < max_k; j++) {
            float a = 0.f;
            uint64_t m = 0;
            if (j < h->site_results.size()) {
                const SiteResult &r = h->site_results[j];
                a = *std::min_element(r.ascores.begin(), r.ascores.end());
                for (uint32_t s : r.positions)
                    if (s >= 1 && s <= 64) m |= 1ull << (s - 1);
            }
            ascores[i * max_k + j] = a;
            alt_mask[i * max_k + j] = m;
        }
    }
    return 0;
}

uint64_t orc_binned(orc_handle *h, double *mz, double *inten, int32_t *bin, int32_t *rank,
                    uint64_t cap, float *min_mz, float *max_mz, uint64_t *n_bins) {
    if (min_mz) *min_mz = h->spec.min_mz;
    if (max_mz) *max_mz = h->spec.max_mz;
    if (n_bins) *n_bins = h->spec.n_bins;
    uint64_t n = 0;
    for (size_t b = 0; b < h->spec.bins.size(); b++)
        for (size_t r = 0; r < h->spec.bins[b].size(); r++) {
            if (n < cap) {
                mz[n] = h->spec.bins[b][r].mz;
                inten[n] = h->spec.bins[b][r].intensity;
                bin[n] = (int32_t)b;
                rank[n] = (int32_t)r;
            }
            n++;
        }
    return n;
}

uint64_t orc_fragments(orc_handle *h, char type, uint64_t charge, const int32_t *sig, float *mz,
                       int32_t *frag_size, int32_t *is_loss, uint64_t cap) {
    if (!known_type(type)) return (uint64_t)-1;
    std::vector<Fragment> f;
    walk(h->pep, residue_flags(h->pep, sig_to_bits(sig, h->pep.n_sites())), type, charge, f);
    for (size_t i = 0; i < f.size() && i < cap; i++) {
        mz[i] = f[i].mz;
        if (frag_size) frag_size[i] = f[i].size;
        if (is_loss) is_loss[i] = f[i].loss_pos > 0;
    }
    return f.size();
}

uint64_t orc_signature_order(orc_handle *h, char type, int32_t *sig, uint64_t cap_rows) {
    if (!known_type(type)) return (uint64_t)-1;
    size_t n = h->pep.n_sites();
    auto combos = combinations(n, h->pep.n_of_mod);
    for (size_t i = 0; i < combos.size() && i < cap_rows; i++) {
        uint64_t bits = combo_bits(combos[i], n, forward_type(type));
        for (size_t j = 0; j < n; j++) sig[i * n + j] = (int32_t)(bits >> j & 1);
    }
    return combos.size();
}

uint64_t orc_site_determining(orc_handle *h, const int32_t *sig1, const int32_t *sig2, char type,
                              uint64_t max_charge, float *out1, uint64_t *n1, float *out2,
                              uint64_t *n2, uint64_t cap) {
    std::vector<float> ions[2];
    size_t n = h->pep.n_sites();
    h->site_determining(sig_to_bits(sig1, n), sig_to_bits(sig2, n), type, max_charge, ions);
    *n1 = ions[0].size();
    *n2 = ions[1].size();
    for (size_t i = 0; i < ions[0].size() && i < cap; i++) out1[i] = ions[0][i];
    for (size_t i = 0; i < ions[1].size() && i < cap; i++) out2[i] = ions[1][i];
    return ions[0].size() + ions[1].size();
}

uint64_t orc_get_peptide(orc_handle *h, const int32_t *sig, uint64_t sig_len, char *buf,
                         uint64_t cap) {
    std::vector<int> s(sig, sig + sig_len);
    return copy_str(format_peptide(h->pep, h->mod_group, h->mod_mass, s), buf, cap);
}

float orc_log_sum(float a, float b) { return log_sum(a, b); }
float orc_log_bin_coef(uint64_t k, uint64_t n) { return log_bin_coef(k, n); }
float orc_binom_log_pmf(float p, uint64_t k, uint64_t n) { return Binom(p).log_pmf(k, n); }
float orc_binom_log_pvalue(float p, uint64_t k, uint64_t n) {
    if (k > n) return -1.f;
    return Binom(p).tail(n)[k];
}
float orc_binom_log10_pvalue(float p, uint64_t k, uint64_t n) {
    if (k > n) return -1.f;
    return log10_of(Binom(p).tail(n)[k]);
}
uint64_t orc_power_set_sums(const float *target, uint64_t n, uint64_t max_depth, float *out,
                            uint64_t cap) {
    std::vector<float> t(target, target + n);
    std::vector<float> s = power_set_sums(t, max_depth);
    for (size_t i = 0; i < s.size() && i < cap; i++) out[i] = s[i];
    return s.size();
}

} /* extern "C" */

/* The reference's sort call itself (cpp/Ascore.cpp:141-146) on caller keys: used to check the
 * on-device emulation (pya_debug_sort).  perm[r] = original index of the r-th sorted element. */
extern "C" void orc_std_sort(const float *keys, uint64_t n, uint32_t *perm) {
    struct Rec {
        float weighted_score;
        uint32_t idx;
    };
    std::vector<Rec> v(n);
    for (uint64_t i = 0; i < n; i++) v[i] = {keys[i], (uint32_t)i};
    std::sort(v.begin(), v.end(),
              [](const Rec &a, const Rec &b) { return a.weighted_score > b.weighted_score; });
    for (uint64_t i = 0; i < n; i++) perm[i] = v[i].idx;
}
