/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C-ABI shim (oracle_abi.h) over the reference's own C++ core.  The reference sources are
 * NOT copied: oracle/Makefile compiles them where they lie under /root/reference and links
 * this file against them into oracle/_ref/libascore_ref.so.  This file only re-states the
 * orchestration the reference keeps in Cython (Ascore.pyx:129-152: bin the spectrum, build
 * the peptide, feed every retained peak to consumePeak in (bin, rank) order, call
 * Ascore::score) because Cython output is generated code and is not built here.
 */
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#include "Ascore.h"
#include "ModifiedPeptide.h"
#include "Spectra.h"
#include "Util.h"
#include "oracle_abi.h"

using namespace ptmscoring;

struct orc_handle {
    BinnedSpectra *spec;
    ModifiedPeptide *pep;
    Ascore *asc;
    size_t n_top;
    orc_handle(float bin_size, size_t n_top_, const char *mg, float mm, float err, const char *ft)
        : spec(new BinnedSpectra(bin_size, n_top_)),
          pep(new ModifiedPeptide(mg, mm, err, ft)),
          asc(new Ascore()),
          n_top(n_top_) {}
    ~orc_handle() {
        delete spec;
        delete pep;
        delete asc;
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

static std::vector<size_t> to_sig(const int32_t *sig, uint64_t n) {
    std::vector<size_t> v(n);
    for (uint64_t i = 0; i < n; i++) v[i] = (size_t)sig[i];
    return v;
}

/* Ascore.pyx:142-150 */
static void feed_peaks(orc_handle *h) {
    BinnedSpectra &s = *h->spec;
    s.resetBin();
    while (s.getBin() < s.getNBins()) {
        s.resetRank();
        while (s.getRank() < s.getNPeaks()) {
            h->pep->consumePeak(s.getMZ(), s.getRank());
            s.nextRank();
        }
        s.nextBin();
    }
}

extern "C" {

const char *orc_impl_name(void) { return "reference"; }

orc_handle *orc_create(float bin_size, uint64_t n_top, const char *mod_group, float mod_mass,
                       float mz_error, const char *fragment_types) {
    return new orc_handle(bin_size, n_top, mod_group, mod_mass, mz_error, fragment_types);
}
void orc_destroy(orc_handle *h) { delete h; }

void orc_add_neutral_loss(orc_handle *h, const char *group, float mass) {
    h->pep->addNeutralLoss(group, mass);
}

int orc_consume_spectra(orc_handle *h, const double *mz, const double *inten, uint64_t n_peaks) {
    try {
        h->spec->consumeSpectra(mz, inten, n_peaks);
    } catch (...) {
        return -1;
    }
    return 0;
}

int orc_consume_peptide(orc_handle *h, const char *peptide, uint64_t n_of_mod,
                        uint64_t max_fragment_charge, const uint32_t *aux_pos,
                        const float *aux_mass, uint64_t n_aux) {
    try {
        if (aux_pos && aux_mass)
            h->pep->consumePeptide(peptide, n_of_mod, max_fragment_charge, aux_pos, aux_mass, n_aux);
        else
            h->pep->consumePeptide(peptide, n_of_mod, max_fragment_charge);
    } catch (...) {
        return -1;
    }
    return 0;
}

int orc_score(orc_handle *h, const double *mz, const double *inten, uint64_t n_peaks,
              const char *peptide, uint64_t n_of_mod, uint64_t max_fragment_charge,
              const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux) {
    try {
        h->spec->consumeSpectra(mz, inten, n_peaks);
        if (aux_pos && aux_mass)
            h->pep->consumePeptide(peptide, n_of_mod, max_fragment_charge, aux_pos, aux_mass, n_aux);
        else
            h->pep->consumePeptide(peptide, n_of_mod, max_fragment_charge);
        feed_peaks(h);
        h->asc->score(*h->spec, *h->pep);
    } catch (...) {
        return -1;
    }
    return 0;
}

uint64_t orc_n_pep_scores(orc_handle *h) { return h->asc->getAllPepScores().size(); }
uint64_t orc_sig_len(orc_handle *h) { return h->pep->getNumberModifiable(); }
uint64_t orc_n_top(orc_handle *h) { return h->n_top; }

void orc_get_pep_scores(orc_handle *h, int32_t *sig, int32_t *counts, float *scores, float *ws,
                        int64_t *nfrag) {
    std::vector<ScoreContainer> v = h->asc->getAllPepScores();
    for (size_t i = 0; i < v.size(); i++) {
        const ScoreContainer &c = v[i];
        size_t sl = c.signature.size();
        for (size_t j = 0; j < sl; j++) sig[i * sl + j] = (int32_t)c.signature[j];
        for (size_t j = 0; j < c.counts.size(); j++) counts[i * c.counts.size() + j] = (int32_t)c.counts[j];
        for (size_t j = 0; j < c.scores.size(); j++) scores[i * c.scores.size() + j] = c.scores[j];
        ws[i] = c.weighted_score;
        nfrag[i] = (int64_t)c.total_fragments;
    }
}

float orc_best_score(orc_handle *h) { return h->asc->getBestScore(); }
uint64_t orc_best_sequence(orc_handle *h, char *buf, uint64_t cap) {
    return copy_str(h->asc->getBestSequence(), buf, cap);
}
uint64_t orc_sequence(orc_handle *h, uint64_t idx, char *buf, uint64_t cap) {
    std::vector<std::string> v = h->asc->getAllSequences();
    if (idx >= v.size()) return 0;
    return copy_str(v[idx], buf, cap);
}
uint64_t orc_n_ascores(orc_handle *h) { return h->asc->getAscores().size(); }
void orc_get_ascores(orc_handle *h, float *out) {
    std::vector<float> v = h->asc->getAscores();
    std::copy(v.begin(), v.end(), out);
}
uint64_t orc_alt_sites(orc_handle *h, uint64_t site, uint32_t *buf, uint64_t cap) {
    std::vector<size_t> v = h->asc->getAlternativeSites(site);
    for (size_t i = 0; i < v.size() && i < cap; i++) buf[i] = (uint32_t)v[i];
    return v.size();
}

float orc_calculate_ambiguity(orc_handle *h, const int32_t *sig_ref, const float *scores_ref,
                              float ws_ref, const int32_t *sig_other, const float *scores_other,
                              float ws_other, uint64_t sig_len, uint64_t n_scores) {
    /* Ascore.pyx:183-206 */
    ScoreContainer a, b;
    a.signature = to_sig(sig_ref, sig_len);
    b.signature = to_sig(sig_other, sig_len);
    a.scores.assign(scores_ref, scores_ref + n_scores);
    b.scores.assign(scores_other, scores_other + n_scores);
    a.counts.assign(n_scores, 0);
    b.counts.assign(n_scores, 0);
    a.weighted_score = ws_ref;
    b.weighted_score = ws_other;
    a.total_fragments = b.total_fragments = 0;
    return h->asc->calculateAmbiguity(a, b);
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
        best_score[i] = h->asc->getBestScore();
        std::vector<ScoreContainer> v = h->asc->getAllPepScores();
        n_sig[i] = (int32_t)v.size();
        uint64_t bits = 0;
        if (!v.empty())
            for (size_t j = 0; j < v[0].signature.size() && j < 64; j++)
                if (v[0].signature[j]) bits |= (1ull << j);
        best_sig[i] = bits;
        std::vector<float> a = h->asc->getAscores();
        for (uint64_t j = 0; j < max_k; j++) {
            ascores[i * max_k + j] = j < a.size() ? a[j] : 0.f;
            uint64_t m = 0;
            if (j < a.size()) {
                for (size_t s : h->asc->getAlternativeSites(j))
                    if (s >= 1 && s <= 64) m |= (1ull << (s - 1));
            }
            alt_mask[i * max_k + j] = m;
        }
    }
    return 0;
}

uint64_t orc_binned(orc_handle *h, double *mz, double *inten, int32_t *bin, int32_t *rank,
                    uint64_t cap, float *min_mz, float *max_mz, uint64_t *n_bins) {
    BinnedSpectra &s = *h->spec;
    if (min_mz) *min_mz = s.getMinMZ();
    if (max_mz) *max_mz = s.getMaxMZ();
    if (n_bins) *n_bins = s.getNBins();
    uint64_t n = 0;
    s.resetBin();
    while (s.getBin() < s.getNBins()) {
        s.resetRank();
        while (s.getRank() < s.getNPeaks()) {
            if (n < cap) {
                mz[n] = s.getMZ();
                inten[n] = s.getIntensity();
                bin[n] = (int32_t)s.getBin();
                rank[n] = (int32_t)s.getRank();
            }
            n++;
            s.nextRank();
        }
        s.nextBin();
    }
    s.resetBin();
    s.resetRank();
    return n;
}

uint64_t orc_fragments(orc_handle *h, char type, uint64_t charge, const int32_t *sig, float *mz,
                       int32_t *frag_size, int32_t *is_loss, uint64_t cap) {
    uint64_t n = 0;
    try {
        ModifiedPeptide::FragmentGraph g = h->pep->getFragmentGraph(type, charge);
        g.setSignature(to_sig(sig, h->pep->getNumberModifiable()));
        for (; !g.isFragmentEnd(); g.incrFragment()) {
            if (n < cap) {
                mz[n] = g.getFragmentMZ();
                if (frag_size) frag_size[n] = (int32_t)g.getFragmentSize();
                if (is_loss) is_loss[n] = g.isLoss() ? 1 : 0;
            }
            n++;
        }
    } catch (...) {
        return (uint64_t)-1;
    }
    return n;
}

uint64_t orc_signature_order(orc_handle *h, char type, int32_t *sig, uint64_t cap_rows) {
    uint64_t n = 0;
    size_t sl = h->pep->getNumberModifiable();
    try {
        for (ModifiedPeptide::FragmentGraph g = h->pep->getFragmentGraph(type, 1);
             !g.isSignatureEnd(); g.incrSignature()) {
            if (n < cap_rows) {
                std::vector<size_t> s = g.getSignature();
                for (size_t j = 0; j < sl; j++) sig[n * sl + j] = (int32_t)s[j];
            }
            n++;
        }
    } catch (...) {
        return (uint64_t)-1;
    }
    return n;
}

uint64_t orc_site_determining(orc_handle *h, const int32_t *sig1, const int32_t *sig2, char type,
                              uint64_t max_charge, float *out1, uint64_t *n1, float *out2,
                              uint64_t *n2, uint64_t cap) {
    size_t sl = h->pep->getNumberModifiable();
    std::vector<std::vector<float>> ions =
        h->pep->getSiteDeterminingIons(to_sig(sig1, sl), to_sig(sig2, sl), type, max_charge);
    *n1 = ions[0].size();
    *n2 = ions[1].size();
    for (size_t i = 0; i < ions[0].size() && i < cap; i++) out1[i] = ions[0][i];
    for (size_t i = 0; i < ions[1].size() && i < cap; i++) out2[i] = ions[1][i];
    return ions[0].size() + ions[1].size();
}

uint64_t orc_get_peptide(orc_handle *h, const int32_t *sig, uint64_t sig_len, char *buf,
                         uint64_t cap) {
    return copy_str(h->pep->getPeptide(to_sig(sig, sig_len)), buf, cap);
}

float orc_log_sum(float a, float b) { return LogMath().log_sum(a, b); }
float orc_log_bin_coef(uint64_t k, uint64_t n) { return LogMath().log_bin_coef(k, n); }
float orc_binom_log_pmf(float p, uint64_t k, uint64_t n) { return BinomialDist(p).log_pmf(k, n); }
float orc_binom_log_pvalue(float p, uint64_t k, uint64_t n) {
    try {
        return BinomialDist(p).log_pvalue(k, n);
    } catch (...) {
        return -1.f;
    }
}
float orc_binom_log10_pvalue(float p, uint64_t k, uint64_t n) {
    try {
        return BinomialDist(p).log10_pvalue(k, n);
    } catch (...) {
        return -1.f;
    }
}
uint64_t orc_power_set_sums(const float *target, uint64_t n, uint64_t max_depth, float *out,
                            uint64_t cap) {
    std::vector<float> t(target, target + n);
    PowerSetSum p(t, max_depth);
    uint64_t c = 0;
    for (;;) {
        if (c < cap) out[c] = p.getSum();
        c++;
        if (!p.hasNext()) break;
        p.next();
    }
    return c;
}

} /* extern "C" */
