/* aux_api.cpp -- host side of the auxiliary scripting classes of pyAscore's ptm_scoring module
 * (SURVEY.md 8(f)-1): the C ABI behind pyascore_amd.PyBinnedSpectra, PyModifiedPeptide,
 * PyFragmentGraph, PyLogMath, PyBinomialDist and PyPowerSetSum (include/pyascore_aux.h).
 *
 * These classes are not on the GPU path -- PyAscore.score never calls them -- they expose single
 * steps of the algorithm (one spectrum's window table, one peptide's fragment walk, one binomial
 * tail) to scripts and to the reference's known-answer unit tests.  One object handles one
 * spectrum or peptide at a time, so they run on the host; every arithmetic step that the kernels
 * also perform (float32 running sums, the double-precision ion offsets, the float32 binomial
 * chain) is written in the same operation order here, and tests/test_aux_api.py cross-checks them
 * against what the kernels count.
 *
 * Behaviour restated from the reference (citations into pyascore/ptm_scoring/):
 *   window table + cursor     cpp/Spectra.cpp:24-107
 *   residues / fixed mods / neutral losses   cpp/ModifiedPeptide.cpp:24-79, 99-124
 *   match cache               cpp/ModifiedPeptide.cpp:126-150 (net semantics)
 *   peptide string            cpp/ModifiedPeptide.cpp:199-253
 *   site-determining ions     cpp/ModifiedPeptide.cpp:259-320
 *   fragment walker           cpp/ModifiedPeptide.cpp:326-609
 *   log math / binomial / subset sums        cpp/Util.cpp:16-160
 * Where the reference throws an int (-> std::terminate under Cython) these functions return
 * PYA_ERR_ARG / PYA_ERR_STATE and the Python classes raise.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pyascore_aux.h"
#include "binom_chain.h"

namespace {

const double kWater = 18.010565, kAmmonia = 17.026549, kAmine = 16.018724, kProton = 1.007825;

float residue_mass_of(char c) {                            /* Types.h:7-30 */
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
    return 0.f;
}

bool travels_forward(char t) { return t == 'b' || t == 'c'; }
bool known_ion_type(char t) { return t == 'b' || t == 'c' || t == 'y' || t == 'z' || t == 'Z'; }

/* sorted, exactly-deduplicated sums of at most `depth` elements of t (0 = no limit), with 0 first */
std::vector<float> subset_sums(const std::vector<float> &t, size_t depth) {
    if (depth > t.size()) depth = t.size();
    const size_t limit = depth == 0 ? (size_t)-1 : depth;  /* Util.cpp:100: `depth < max_depth - 1` wraps at 0 */
    std::vector<float> out{0.f};
    struct Frame { size_t next; float base; size_t level; };
    std::vector<Frame> todo{{0, 0.f, 0}};
    while (!todo.empty()) {
        Frame f = todo.back();
        todo.pop_back();
        for (size_t i = f.next; i < t.size(); i++) {
            const float s = f.base + t[i];
            out.push_back(s);
            if (i + 1 < t.size() && f.level + 1 < limit) todo.push_back({i + 1, s, f.level + 1});
        }
    }
    std::sort(out.begin(), out.end());
    out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
}

uint64_t copy_out(const std::string &s, char *buf, uint64_t cap) {
    if (buf && cap) {
        const size_t n = std::min<size_t>(s.size(), cap - 1);
        std::memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return s.size();
}

}  // namespace

/* ---------------------------------------------------------------------------------------------- */
/* window table                                                                                   */
/* ---------------------------------------------------------------------------------------------- */
struct pya_spectra {
    struct Pk { double mz, inten; };
    float bin_size = 100.f, min_mz = 0.f, max_mz = INFINITY;
    uint64_t n_top = 10, n_bins = ~0ull;                   /* Spectra.cpp:15-17: -1 before the first spectrum */
    std::vector<std::vector<Pk>> windows;
};

/* ---------------------------------------------------------------------------------------------- */
/* modified peptide                                                                               */
/* ---------------------------------------------------------------------------------------------- */
struct pya_modpep {
    std::string mod_group, fragment_types;
    float mod_mass = 0.f, mz_error = 0.5f;
    std::map<char, float> loss_of;                         /* letter -> neutral loss */
    /* consumed peptide */
    std::string seq;
    uint64_t n_of_mod = 0, max_charge = 1;
    std::vector<uint32_t> aux_pos;
    std::vector<float> aux_mass;
    struct Res {
        float mass[2];                                     /* unmodified, modified */
        float loss[2];
        bool modifiable;
        bool loss1_dropped;                                /* fixed mod on a variable site: the reference */
    };                                                     /* shrinks the loss vector to one entry        */
    std::vector<Res> res;
    std::vector<uint32_t> sites;                           /* residue index of every modifiable residue, N -> C */
    /* match cache: retained peaks as consume_peak received them */
    struct Hit { float mz; uint64_t rank; };
    std::vector<Hit> peaks;
    bool have_peptide = false;
    std::string err;

    int fail(int code, const char *msg) {
        err = msg;
        return code;
    }
    bool site_letter(char c, size_t i, size_t L) const {
        return mod_group.find(c) != std::string::npos || (i == 0 && mod_group.find('n') != std::string::npos) ||
               (i + 1 == L && mod_group.find('c') != std::string::npos);
    }
    /* min rank over consumed peaks p with f32(f - err) < p < f32(f + err) and f >= p - 0.5 */
    bool match(float f, float *mz, uint64_t *rank) const {
        const float lo = f - mz_error, hi = f + mz_error;
        bool found = false;
        for (const Hit &h : peaks) {
            if (!(h.mz > lo && h.mz < hi)) continue;
            if ((double)f < (double)h.mz - .5) continue;
            if (!found || h.rank < *rank) {
                *rank = h.rank;
                *mz = h.mz;
                found = true;
            }
        }
        return found;
    }
};

/* m/z of one fragment: float (sum - loss), ion offsets one at a time in double, charge, narrowed */
static float ion_mz(float running, float loss, char type, uint64_t charge) {
    double m = (double)(running - loss);
    if (type == 'y') {
        m += kWater;
    } else if (type == 'z') {
        m += kWater;
        m -= kAmmonia;
    } else if (type == 'Z') {
        m += kWater;
        m -= kAmine;
    } else if (type == 'c') {
        m += kAmmonia;
    }
    if (charge > 0) m = (m + (double)charge * kProton) / (double)charge;
    return (float)m;
}

/* ---------------------------------------------------------------------------------------------- */
/* fragment walker: one ion type and charge, all site assignments in the reference's order        */
/* ---------------------------------------------------------------------------------------------- */
struct pya_fgraph {
    const pya_modpep *pep;
    char type;
    uint64_t charge;
    bool forward;
    size_t L = 0;
    /* site assignment: flag per modifiable residue in TRAVEL order */
    std::vector<uint32_t> site_res;                        /* residue index, travel order */
    std::vector<uint8_t> site_on;
    uint64_t outstanding = 0;                              /* > 0 <=> past the last assignment */
    /* walk state: `at` = residues travelled before the current one (L = past the end) */
    size_t at = 0;
    std::vector<float> prefix;                             /* running float32 sums, one per residue travelled */
    std::string letters;
    std::vector<size_t> losses_upto;                       /* size of the loss stack after each residue */
    std::vector<float> loss_stack;
    std::vector<float> variants{0.f};                      /* subset_sums(loss_stack, 2) */
    size_t variant = 0;

    size_t residue_at(size_t d) const { return forward ? d : L - 1 - d; }
    size_t distance_of(size_t residue) const { return forward ? residue : L - 1 - residue; }
    int state_of(size_t residue) const {
        for (size_t j = 0; j < site_res.size(); j++)
            if (site_res[j] == residue) return site_on[j];
        return 0;
    }
    /* the current residue joins the fragment (ModifiedPeptide.cpp:379-408) */
    void take_residue() {
        const size_t r = residue_at(at);
        const int st = state_of(r);
        const pya_modpep::Res &x = pep->res[r];
        float v = x.mass[st];
        if (!prefix.empty()) v += prefix.back();
        prefix.push_back(v);
        letters.push_back(pep->seq[r]);
        const float loss = (st == 1 && x.loss1_dropped) ? 0.f : x.loss[st];
        if (loss != 0.f) {
            loss_stack.push_back(loss);
            variants = subset_sums(loss_stack, 2);
        }
        losses_upto.push_back(loss_stack.size());
        variant = 0;
    }
    void restart_walk() {
        prefix.clear();
        letters.clear();
        losses_upto.clear();
        loss_stack.clear();
        variants.assign(1, 0.f);
        variant = 0;
        at = 0;
        take_residue();
    }
    void first_assignment() {
        L = pep->seq.size();
        site_res.clear();
        site_on.clear();
        outstanding = pep->n_of_mod;
        for (size_t d = 0; d < L; d++) {
            const size_t r = residue_at(d);
            if (!pep->res[r].modifiable) continue;
            site_res.push_back((uint32_t)r);
            site_on.push_back(outstanding ? 1 : 0);
            if (outstanding) outstanding--;
        }
        restart_walk();
    }
    bool more_variants() const { return variant + 1 < variants.size(); }
    bool walk_done() const { return at == L - 1 && !more_variants(); }
};

extern "C" {

/* ---- PyBinnedSpectra ---- */
pya_spectra *pya_spectra_create(float bin_size, uint64_t n_top) {
    pya_spectra *s = new pya_spectra;
    s->bin_size = bin_size;
    s->n_top = n_top;
    return s;
}
void pya_spectra_destroy(pya_spectra *s) { delete s; }

int pya_spectra_consume(pya_spectra *s, const double *mz, const double *inten, uint64_t n) {
    if (!s || !mz || !inten || n == 0) return PYA_ERR_ARG;
    const double lo = *std::min_element(mz, mz + n), hi = *std::max_element(mz, mz + n);
    s->min_mz = (float)(std::floor(lo / 100.) * 100.);     /* the 100 is fixed, whatever bin_size is */
    s->max_mz = (float)(std::ceil(hi / 100.) * 100.);
    const float span = (s->max_mz - s->min_mz) / s->bin_size;
    s->n_bins = (uint64_t)std::ceil(span);
    s->windows.assign(s->n_bins, {});
    if (s->n_bins == 0) return PYA_ERR_PSM;                /* reference: out-of-bounds write */
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t w = (uint64_t)std::floor((mz[i] - (double)s->min_mz) / (double)s->bin_size);
        s->windows[std::min(w, s->n_bins - 1)].push_back({mz[i], inten[i]});
    }
    auto brighter = [](const pya_spectra::Pk &a, const pya_spectra::Pk &b) { return a.inten > b.inten; };
    for (auto &w : s->windows) {
        if (s->n_top < w.size()) {
            if (s->n_top > 0) std::nth_element(w.begin(), w.begin() + (s->n_top - 1), w.end(), brighter);
            w.resize(s->n_top);
        }
        std::sort(w.begin(), w.end(), brighter);
    }
    return PYA_OK;
}

void pya_spectra_info(const pya_spectra *s, float *min_mz, float *max_mz, float *bin_size, uint64_t *n_bins,
                      uint64_t *n_top) {
    if (!s) return;
    if (min_mz) *min_mz = s->min_mz;
    if (max_mz) *max_mz = s->max_mz;
    if (bin_size) *bin_size = s->bin_size;
    if (n_bins) *n_bins = s->n_bins;
    if (n_top) *n_top = s->n_top;
}

int64_t pya_spectra_window_size(const pya_spectra *s, uint64_t window) {
    if (!s || window >= s->windows.size()) return -1;      /* reference: std::out_of_range */
    return (int64_t)s->windows[window].size();
}

int pya_spectra_peak(const pya_spectra *s, uint64_t window, uint64_t rank, double *mz, double *inten) {
    if (!s || window >= s->windows.size() || rank >= s->windows[window].size()) return PYA_ERR_ARG;
    if (mz) *mz = s->windows[window][rank].mz;
    if (inten) *inten = s->windows[window][rank].inten;
    return PYA_OK;
}

/* ---- PyModifiedPeptide ---- */
pya_modpep *pya_modpep_create(const char *mod_group, float mod_mass, float mz_error, const char *fragment_types) {
    if (!mod_group || !fragment_types) return nullptr;
    pya_modpep *p = new pya_modpep;
    p->mod_group = mod_group;
    p->fragment_types = fragment_types;
    p->mod_mass = mod_mass;
    p->mz_error = mz_error;
    return p;
}
void pya_modpep_destroy(pya_modpep *p) { delete p; }
const char *pya_modpep_last_error(const pya_modpep *p) { return p ? p->err.c_str() : "NULL peptide object"; }

int pya_modpep_add_neutral_loss(pya_modpep *p, const char *group, float mass) {
    if (!p || !group) return PYA_ERR_ARG;
    for (const char *c = group; *c; c++) p->loss_of[*c] = mass;
    return PYA_OK;
}

int pya_modpep_consume_peptide(pya_modpep *p, const char *peptide, uint64_t len, uint64_t n_of_mod,
                               uint64_t max_charge, const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux) {
    if (!p || !peptide) return PYA_ERR_ARG;
    if (len == 0) return p->fail(PYA_ERR_PSM, "empty peptide");
    if (n_aux && (!aux_pos || !aux_mass)) return p->fail(PYA_ERR_ARG, "NULL fixed-modification arrays");
    std::vector<pya_modpep::Res> res(len);
    for (uint64_t i = 0; i < len; i++) {
        const char c = peptide[i];
        const float m = residue_mass_of(c);
        if (m == 0.f) {
            char msg[96];
            std::snprintf(msg, sizeof msg, "unknown residue '%c' at position %llu", c, (unsigned long long)(i + 1));
            return p->fail(PYA_ERR_PSM, msg);
        }
        pya_modpep::Res &r = res[i];
        r.mass[0] = m;
        r.mass[1] = 0.f;
        r.loss[0] = r.loss[1] = 0.f;
        r.loss1_dropped = false;
        auto up = p->loss_of.find(c);                       /* upper case: the unmodified residue loses it */
        if (up != p->loss_of.end()) r.loss[0] = up->second;
        r.modifiable = p->site_letter(c, (size_t)i, (size_t)len);
        if (r.modifiable) {
            r.mass[1] = m + p->mod_mass;
            auto low = p->loss_of.find((char)std::tolower(c));   /* lower case: the modified residue */
            if (low != p->loss_of.end()) r.loss[1] = low->second;
        }
    }
    for (uint64_t a = 0; a < n_aux; a++) {                  /* fixed modifications: position 0 = n-terminus */
        const uint64_t i = aux_pos[a] > 0 ? aux_pos[a] - 1 : 0;
        if (i >= len) return p->fail(PYA_ERR_PSM, "aux_mod_pos beyond the peptide");
        res[i].mass[0] += aux_mass[a];
        if (res[i].modifiable) res[i].mass[1] += aux_mass[a];
        auto low = p->loss_of.find((char)std::tolower(peptide[i]));
        if (low != p->loss_of.end()) {                      /* a fixed-modified residue takes the lower-case loss */
            res[i].loss[0] = low->second;
            res[i].loss1_dropped = true;
        }
    }
    p->seq.assign(peptide, len);
    p->n_of_mod = n_of_mod;
    p->max_charge = max_charge;
    p->aux_pos.assign(aux_pos, aux_pos + n_aux);
    p->aux_mass.assign(aux_mass, aux_mass + n_aux);
    p->res.swap(res);
    p->sites.clear();
    for (uint64_t i = 0; i < len; i++)
        if (p->res[i].modifiable) p->sites.push_back((uint32_t)i);
    p->peaks.clear();                                       /* a new peptide starts with an empty match cache */
    p->have_peptide = true;
    return PYA_OK;
}

int64_t pya_modpep_n_modifiable(const pya_modpep *p) { return p && p->have_peptide ? (int64_t)p->sites.size() : -1; }

int pya_modpep_consume_peak(pya_modpep *p, float mz, uint64_t rank) {
    if (!p || !p->have_peptide) return PYA_ERR_STATE;
    p->peaks.push_back({mz, rank});
    return PYA_OK;
}

int pya_modpep_get_match(const pya_modpep *p, float fragment_mz, float *peak_mz, uint64_t *rank) {
    if (!p || !p->have_peptide) return PYA_ERR_STATE;
    float mz = 0.f;
    uint64_t rk = 0;
    const bool hit = p->match(fragment_mz, &mz, &rk);
    if (hit && peak_mz) *peak_mz = mz;
    if (hit && rank) *rank = rk;
    return hit ? 1 : 0;
}

int64_t pya_modpep_get_peptide(const pya_modpep *p, const uint32_t *signature, uint64_t n_sig, char *buf, uint64_t cap) {
    if (!p || !p->have_peptide) return PYA_ERR_STATE;
    const size_t L = p->seq.size(), n = p->sites.size();
    std::vector<uint32_t> sig(signature, signature + (signature ? n_sig : 0));
    if (sig.empty()) sig.assign(std::min<size_t>(p->n_of_mod, n), 1u);   /* default: the first assignment */
    std::vector<float> mm(L + 2, 0.f);                      /* n-terminus, residues, c-terminus */
    const bool has_n = p->mod_group.find('n') != std::string::npos;
    if (p->n_of_mod > n) (has_n ? mm.front() : mm.back()) += p->mod_mass;
    for (size_t j = 0; j < sig.size(); j++) {
        if (sig[j] != 1) continue;
        const size_t pos = j < n ? p->sites[j] : L;
        const char aa = pos < L ? p->seq[pos] : 0;
        if (p->mod_group.find(aa) != std::string::npos) mm[pos + 1] += p->mod_mass;
        else if (pos == 0) mm.front() += p->mod_mass;
        else if (pos + 1 == L) mm.back() += p->mod_mass;
    }
    for (size_t a = 0; a < p->aux_pos.size(); a++)
        if (p->aux_pos[a] < mm.size()) mm[p->aux_pos[a]] += p->aux_mass[a];
    const std::string full = "n" + p->seq + "c";
    std::string out;
    for (size_t i = mm.front() == 0.f ? 1 : 0, e = mm.size() - (mm.back() == 0.f ? 1 : 0); i < e; i++) {
        out += full[i];
        if (mm[i] > 0.f) {
            char t[16];
            std::snprintf(t, sizeof t, "[%d]", (int)std::round(mm[i]));
            out += t;
        }
    }
    return (int64_t)copy_out(out, buf, cap);
}

/* ---- PyFragmentGraph ---- */
pya_fgraph *pya_fgraph_create(const pya_modpep *p, char type, uint64_t charge) {
    if (!p || !p->have_peptide || !known_ion_type(type)) return nullptr;
    pya_fgraph *g = new pya_fgraph;
    g->pep = p;
    g->type = type;
    g->charge = charge;
    g->forward = travels_forward(type);
    g->first_assignment();
    return g;
}
void pya_fgraph_destroy(pya_fgraph *g) { delete g; }
char pya_fgraph_type(const pya_fgraph *g) { return g ? g->type : 0; }
uint64_t pya_fgraph_charge(const pya_fgraph *g) { return g ? g->charge : 0; }

int pya_fgraph_reset_iterator(pya_fgraph *g) {
    if (!g) return PYA_ERR_ARG;
    g->first_assignment();
    return PYA_OK;
}
int pya_fgraph_is_signature_end(const pya_fgraph *g) { return g && g->outstanding > 0 ? 1 : 0; }
int pya_fgraph_is_fragment_end(const pya_fgraph *g) { return g && g->walk_done() ? 1 : 0; }
int pya_fgraph_is_loss(const pya_fgraph *g) { return g && g->variant > 0 ? 1 : 0; }

/* Next assignment = next combination in travel order: the last site that can still move takes one
 * step, every site behind it is packed right after it.  The walk then resumes at the moved site's
 * old position (or where the walk stood, if that is earlier), keeping the shared prefix. */
int pya_fgraph_incr_signature(pya_fgraph *g) {
    if (!g) return PYA_ERR_ARG;
    if (g->outstanding > 0) return PYA_ERR_STATE;           /* reference: throw 40 */
    const size_t M = g->site_on.size();
    uint64_t carry = 1;
    size_t moved_from = g->L;                               /* residue index */
    for (size_t j = M; j-- > 0 && carry;) {
        if (!g->site_on[j]) continue;
        moved_from = g->site_res[j];
        g->site_on[j] = 0;
        if (carry == M - j) {                               /* packed against the end: carried further */
            carry++;
            continue;
        }
        for (size_t t = j + 1; carry; carry--, t++) g->site_on[t]++;
    }
    g->outstanding = carry;
    if (g->outstanding > 0) return PYA_OK;                  /* that was the last assignment */
    const size_t d = g->distance_of(moved_from);
    if (d < g->at) g->at = d;
    g->prefix.resize(g->at);
    g->letters.resize(g->at);
    g->losses_upto.resize(g->at);
    g->loss_stack.resize(g->losses_upto.empty() ? 0 : g->losses_upto.back());
    g->variants = subset_sums(g->loss_stack, 2);
    g->take_residue();
    return PYA_OK;
}

int pya_fgraph_reset_fragment(pya_fgraph *g) {
    if (!g) return PYA_ERR_ARG;
    g->restart_walk();
    return PYA_OK;
}

int pya_fgraph_incr_fragment(pya_fgraph *g) {
    if (!g) return PYA_ERR_ARG;
    if (g->outstanding > 0 || g->walk_done()) return PYA_ERR_STATE;   /* reference: throw 40 */
    if (g->more_variants()) {
        g->variant++;
        return PYA_OK;
    }
    g->at++;
    if (!g->walk_done()) g->take_residue();                 /* the full-length "fragment" is never built */
    return PYA_OK;
}

int pya_fgraph_set_signature(pya_fgraph *g, const uint32_t *signature, uint64_t n) {
    if (!g || (!signature && n)) return PYA_ERR_ARG;
    const size_t M = g->site_on.size();
    if (n != M) return PYA_ERR_ARG;                         /* reference: throw 50 */
    for (size_t j = 0; j < M; j++)                          /* signatures are always given N -> C */
        g->site_on[j] = (uint8_t)signature[g->forward ? j : M - 1 - j];
    g->restart_walk();
    return PYA_OK;
}

int64_t pya_fgraph_get_signature(const pya_fgraph *g, uint64_t *out, uint64_t cap) {
    if (!g) return PYA_ERR_ARG;
    const size_t M = g->site_on.size();
    for (size_t j = 0; j < M && j < cap && out; j++) out[j] = g->site_on[g->forward ? j : M - 1 - j];
    return (int64_t)M;
}

int pya_fgraph_fragment_mz(const pya_fgraph *g, float *mz) {
    if (!g || !mz || g->prefix.empty()) return PYA_ERR_STATE;
    *mz = ion_mz(g->prefix.back(), g->variants[g->variant], g->type, g->charge);
    return PYA_OK;
}
uint64_t pya_fgraph_fragment_size(const pya_fgraph *g) { return g ? g->letters.size() : 0; }
int64_t pya_fgraph_fragment_seq(const pya_fgraph *g, char *buf, uint64_t cap) {
    if (!g) return PYA_ERR_ARG;
    return (int64_t)copy_out(g->letters, buf, cap);
}

/* Site-determining ions of two assignments: every fragment m/z of each (charges 1..max_charge, all
 * neutral-loss variants), sorted; then one pass over both lists drops pairs closer than mz_error
 * and keeps the rest on its own side (ModifiedPeptide.cpp:259-320). */
int pya_modpep_site_ions(const pya_modpep *p, const uint32_t *sig_1, const uint32_t *sig_2, uint64_t n_sig, char type,
                         uint64_t max_charge, float *out_1, uint64_t cap_1, uint64_t *n_1, float *out_2,
                         uint64_t cap_2, uint64_t *n_2) {
    if (!p || !p->have_peptide || !known_ion_type(type) || !n_1 || !n_2) return PYA_ERR_ARG;
    std::vector<float> ions[2];
    const uint32_t *sigs[2] = {sig_1, sig_2};
    for (uint64_t z = 1; z <= max_charge; z++) {
        pya_fgraph g;
        g.pep = p;
        g.type = type;
        g.charge = z;
        g.forward = travels_forward(type);
        g.first_assignment();
        for (int side = 0; side < 2; side++) {
            if (pya_fgraph_set_signature(&g, sigs[side], n_sig) != PYA_OK) return PYA_ERR_ARG;
            while (!g.walk_done()) {
                ions[side].push_back(ion_mz(g.prefix.back(), g.variants[g.variant], type, z));
                pya_fgraph_incr_fragment(&g);
            }
        }
    }
    std::sort(ions[0].begin(), ions[0].end());
    std::sort(ions[1].begin(), ions[1].end());
    std::vector<float> kept[2];
    size_t i = 0, j = 0;
    while (i < ions[0].size() || j < ions[1].size()) {
        if (j == ions[1].size()) kept[0].push_back(ions[0][i++]);
        else if (i == ions[0].size()) kept[1].push_back(ions[1][j++]);
        else if (std::abs(ions[0][i] - ions[1][j]) < p->mz_error) { i++; j++; }
        else if (ions[0][i] < ions[1][j]) kept[0].push_back(ions[0][i++]);
        else kept[1].push_back(ions[1][j++]);
    }
    *n_1 = kept[0].size();
    *n_2 = kept[1].size();
    if (out_1) std::memcpy(out_1, kept[0].data(), std::min<uint64_t>(cap_1, kept[0].size()) * sizeof(float));
    if (out_2) std::memcpy(out_2, kept[1].data(), std::min<uint64_t>(cap_2, kept[1].size()) * sizeof(float));
    return PYA_OK;
}

/* ---- PyLogMath / PyBinomialDist / PyPowerSetSum ---- */
float pya_log_sum(float a, float b) { return pya_chain::log_sum(a, b); }
int pya_log_bin_coef(uint64_t k, uint64_t n, float *out) {
    if (!out || k > n) return PYA_ERR_ARG;
    *out = pya_chain::log_bin_coef(k, n);
    return PYA_OK;
}
int pya_binomial(float prob, int what, uint64_t successes, uint64_t trials, float *out) {
    if (!out || successes > trials) return PYA_ERR_ARG;    /* reference: throw 10 */
    const pya_chain::Binomial d(prob);
    if (what == 0) *out = d.log_pmf(successes, trials);
    else if (what == 1) *out = d.log_pvalue(successes, trials);
    else if (what == 2) *out = pya_chain::log10_of(d.log_pvalue(successes, trials));
    else return PYA_ERR_ARG;
    return PYA_OK;
}
int64_t pya_power_set_sums(const float *target, uint64_t n, uint64_t max_depth, float *out, uint64_t cap) {
    if (n && !target) return PYA_ERR_ARG;
    const std::vector<float> sums = subset_sums(std::vector<float>(target, target + n), (size_t)max_depth);
    if (out) std::memcpy(out, sums.data(), std::min<uint64_t>(cap, sums.size()) * sizeof(float));
    return (int64_t)sums.size();
}

} /* extern "C" */
