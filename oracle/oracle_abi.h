/*
 * oracle_abi.h -- TEST INFRASTRUCTURE ONLY.
 *
 * One C ABI, two implementations:
 *   oracle/_ref/libascore_ref.so     = the reference's own C++ core
 *       (/root/reference/pyascore/ptm_scoring/cpp/{Ascore,ModifiedPeptide,Spectra,Util}.cpp,
 *       compiled where they lie) behind oracle/ref_shim.cpp.
 *   oracle/libascore_oracle.so       = oracle/ascore_oracle.cpp, this repo's CPU
 *       restatement of the same algorithm.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * either library.  The product (pyascore_amd) never does.
 */
#ifndef ORACLE_ABI_H
#define ORACLE_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_handle orc_handle;

/* ---- PyAscore-shaped surface (Ascore.pyx:64-67, :81-99, :103-152, :232-288) ---- */
orc_handle *orc_create(float bin_size, uint64_t n_top, const char *mod_group, float mod_mass,
                       float mz_error, const char *fragment_types);
void orc_destroy(orc_handle *h);
void orc_add_neutral_loss(orc_handle *h, const char *group, float mass);
/* returns 0 on success, <0 if the implementation threw */
int orc_score(orc_handle *h, const double *mz, const double *inten, uint64_t n_peaks,
              const char *peptide, uint64_t n_of_mod, uint64_t max_fragment_charge,
              const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux);

uint64_t orc_n_pep_scores(orc_handle *h);
uint64_t orc_sig_len(orc_handle *h);   /* number of modifiable residues of the last peptide */
uint64_t orc_n_top(orc_handle *h);
/* sig: n x sig_len, counts/scores: n x n_top, ws/nfrag: n   (sorted order of the last score()) */
void orc_get_pep_scores(orc_handle *h, int32_t *sig, int32_t *counts, float *scores, float *ws,
                        int64_t *nfrag);
float orc_best_score(orc_handle *h);
uint64_t orc_best_sequence(orc_handle *h, char *buf, uint64_t cap);
uint64_t orc_sequence(orc_handle *h, uint64_t idx, char *buf, uint64_t cap);
uint64_t orc_n_ascores(orc_handle *h);
void orc_get_ascores(orc_handle *h, float *out);
uint64_t orc_alt_sites(orc_handle *h, uint64_t site, uint32_t *buf, uint64_t cap);
float orc_calculate_ambiguity(orc_handle *h, const int32_t *sig_ref, const float *scores_ref,
                              float ws_ref, const int32_t *sig_other, const float *scores_other,
                              float ws_other, uint64_t sig_len, uint64_t n_scores);

/* Batch driver (CSR inputs) used for the timed CPU baseline and bulk parity.
 * Per PSM it performs exactly one orc_score() and extracts the fixed summary:
 *   best_score[i], best_sig[i] (bit j = site j modified, N-term site = bit 0),
 *   n_sig[i] = number of pep_scores, ascores[i*max_k + j], alt_mask[i*max_k + j]
 *   (bit p-1 = 1-based peptide position p is an alternative site; positions > 64 dropped).
 * Returns 0 or the negative (index+1) of the first PSM that threw. */
int64_t orc_score_batch(orc_handle *h, uint64_t n_psm, const double *mz, const double *inten,
                        const int64_t *peak_off, const char *pep, const int64_t *pep_off,
                        const int32_t *n_of_mod, const int32_t *max_charge,
                        const uint32_t *aux_pos, const float *aux_mass, const int64_t *aux_off,
                        uint64_t max_k, float *best_score, uint64_t *best_sig, int32_t *n_sig,
                        float *ascores, uint64_t *alt_mask);

/* ---- component surface (pins against the reference's known-answer unit tests) ---- */
int orc_consume_spectra(orc_handle *h, const double *mz, const double *inten, uint64_t n_peaks);
/* retained peaks in (bin asc, rank asc) order */
uint64_t orc_binned(orc_handle *h, double *mz, double *inten, int32_t *bin, int32_t *rank,
                    uint64_t cap, float *min_mz, float *max_mz, uint64_t *n_bins);
int orc_consume_peptide(orc_handle *h, const char *peptide, uint64_t n_of_mod,
                        uint64_t max_fragment_charge, const uint32_t *aux_pos,
                        const float *aux_mass, uint64_t n_aux);
/* fragments of ONE signature (N->C order, length sig_len) for (type, charge), walk order */
uint64_t orc_fragments(orc_handle *h, char type, uint64_t charge, const int32_t *sig,
                       float *mz, int32_t *frag_size, int32_t *is_loss, uint64_t cap);
/* signatures in the iteration order of (type) : out n x sig_len, N->C */
uint64_t orc_signature_order(orc_handle *h, char type, int32_t *sig, uint64_t cap_rows);
uint64_t orc_site_determining(orc_handle *h, const int32_t *sig1, const int32_t *sig2, char type,
                              uint64_t max_charge, float *out1, uint64_t *n1, float *out2,
                              uint64_t *n2, uint64_t cap);
uint64_t orc_get_peptide(orc_handle *h, const int32_t *sig, uint64_t sig_len, char *buf,
                         uint64_t cap);

float orc_log_sum(float a, float b);
float orc_log_bin_coef(uint64_t k, uint64_t n);
float orc_binom_log_pmf(float p, uint64_t k, uint64_t n);
float orc_binom_log_pvalue(float p, uint64_t k, uint64_t n);
float orc_binom_log10_pvalue(float p, uint64_t k, uint64_t n);
uint64_t orc_power_set_sums(const float *target, uint64_t n, uint64_t max_depth, float *out,
                            uint64_t cap);

const char *orc_impl_name(void);

#ifdef __cplusplus
}
#endif
#endif
