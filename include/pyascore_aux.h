/*
 * pyascore_aux.h -- C ABI of the auxiliary scripting classes of pyAscore's ptm_scoring module
 * (libpyascore_hip.so).  Host-only: none of these touches the GPU, and
 * PyAscore.score never calls them; they expose single steps of the algorithm to scripts and tests.
 *
 * Every group replaces one Cython wrapper of the reference (citations into pyascore/ptm_scoring/):
 *
 *   pya_spectra_*          PyBinnedSpectra    Spectra.pyx:8-125       (cpp/Spectra.cpp:10-115)
 *   pya_modpep_*           PyModifiedPeptide  ModifiedPeptide.pyx:10-157 (cpp/ModifiedPeptide.cpp:9-320)
 *   pya_fgraph_*           PyFragmentGraph    ModifiedPeptide.pyx:159-329 (cpp/ModifiedPeptide.cpp:326-609)
 *   pya_log_sum / pya_log_bin_coef            PyLogMath      Util.pyx:6-46   (cpp/Util.cpp:16-41)
 *   pya_binomial           PyBinomialDist     Util.pyx:48-98          (cpp/Util.cpp:47-83)
 *   pya_power_set_sums     PyPowerSetSum      Util.pyx:100-134        (cpp/Util.cpp:89-160)
 *
 * Conventions as in pyascore_hip.h: plain pointers and sizes, caller-allocated outputs, status
 * codes (PYA_OK / PYA_ERR_*).  Where the reference throws (and, under Cython, aborts the process)
 * these return PYA_ERR_ARG (bad argument) or PYA_ERR_STATE (call sequence, e.g. stepping past the
 * last fragment).  The cursors of PyBinnedSpectra (bin, rank) and PyPowerSetSum (position) are
 * plain clamped integers and live in the Python classes.
 */
#ifndef PYASCORE_AUX_H
#define PYASCORE_AUX_H

#include <stdint.h>
#include "pyascore_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pya_spectra pya_spectra;
typedef struct pya_modpep pya_modpep;
typedef struct pya_fgraph pya_fgraph;

/* ---- window table of one spectrum ---- */
pya_spectra *pya_spectra_create(float bin_size, uint64_t n_top);
void pya_spectra_destroy(pya_spectra *s);
/* bins the peaks into windows of bin_size and keeps the n_top most intense of each, most intense
 * first (ties as std::nth_element + std::sort leave them) */
int pya_spectra_consume(pya_spectra *s, const double *mz, const double *intensity, uint64_t n);
void pya_spectra_info(const pya_spectra *s, float *min_mz, float *max_mz, float *bin_size, uint64_t *n_bins,
                      uint64_t *n_top);
int64_t pya_spectra_window_size(const pya_spectra *s, uint64_t window);   /* -1: no such window */
int pya_spectra_peak(const pya_spectra *s, uint64_t window, uint64_t rank, double *mz, double *intensity);

/* ---- one modified peptide ---- */
pya_modpep *pya_modpep_create(const char *mod_group, float mod_mass, float mz_error, const char *fragment_types);
void pya_modpep_destroy(pya_modpep *p);
const char *pya_modpep_last_error(const pya_modpep *p);
int pya_modpep_add_neutral_loss(pya_modpep *p, const char *group, float mass);
int pya_modpep_consume_peptide(pya_modpep *p, const char *peptide, uint64_t len, uint64_t n_of_mod,
                               uint64_t max_fragment_charge, const uint32_t *aux_mod_pos,
                               const float *aux_mod_mass, uint64_t n_aux);
int64_t pya_modpep_n_modifiable(const pya_modpep *p);
/* the match cache: peaks as (float m/z, rank inside the window); get_match returns 1 and the
 * lowest-ranked peak within mz_error of the theoretical m/z, 0 if there is none */
int pya_modpep_consume_peak(pya_modpep *p, float mz, uint64_t rank);
int pya_modpep_get_match(const pya_modpep *p, float fragment_mz, float *peak_mz, uint64_t *rank);
/* "PEPT[80]IDEK"; signature = one 0/1 per modifiable residue, N -> C (NULL: the first assignment).
 * Returns the string length (truncated to cap - 1 in buf) */
int64_t pya_modpep_get_peptide(const pya_modpep *p, const uint32_t *signature, uint64_t n_sig, char *buf,
                               uint64_t cap);
/* fragments of one assignment that have no partner within mz_error among the other's, charges
 * 1..max_charge; *n_1 / *n_2 receive the counts (call with cap 0 to size the arrays) */
int pya_modpep_site_ions(const pya_modpep *p, const uint32_t *sig_1, const uint32_t *sig_2, uint64_t n_sig,
                         char fragment_type, uint64_t max_charge, float *out_1, uint64_t cap_1, uint64_t *n_1,
                         float *out_2, uint64_t cap_2, uint64_t *n_2);

/* ---- fragment walker over the site assignments of a peptide (which must outlive it) ---- */
pya_fgraph *pya_fgraph_create(const pya_modpep *p, char fragment_type, uint64_t charge_state);
void pya_fgraph_destroy(pya_fgraph *g);
char pya_fgraph_type(const pya_fgraph *g);
uint64_t pya_fgraph_charge(const pya_fgraph *g);
int pya_fgraph_reset_iterator(pya_fgraph *g);
int pya_fgraph_incr_signature(pya_fgraph *g);
int pya_fgraph_is_signature_end(const pya_fgraph *g);
int pya_fgraph_reset_fragment(pya_fgraph *g);
int pya_fgraph_incr_fragment(pya_fgraph *g);
int pya_fgraph_is_fragment_end(const pya_fgraph *g);
int pya_fgraph_is_loss(const pya_fgraph *g);
int pya_fgraph_set_signature(pya_fgraph *g, const uint32_t *signature, uint64_t n);
int64_t pya_fgraph_get_signature(const pya_fgraph *g, uint64_t *out, uint64_t cap);   /* returns the length */
int pya_fgraph_fragment_mz(const pya_fgraph *g, float *mz);
uint64_t pya_fgraph_fragment_size(const pya_fgraph *g);
int64_t pya_fgraph_fragment_seq(const pya_fgraph *g, char *buf, uint64_t cap);

/* ---- float32 log-space arithmetic of the scores ---- */
float pya_log_sum(float a, float b);
int pya_log_bin_coef(uint64_t k, uint64_t n, float *out);
/* what: 0 log_pmf, 1 log_pvalue (upper tail, inclusive), 2 log10_pvalue */
int pya_binomial(float prob, int what, uint64_t successes, uint64_t trials, float *out);
/* {0} and every sum of at most max_depth elements of target (0 = no limit), ascending, exact
 * duplicates removed; returns the count (fills at most cap) */
int64_t pya_power_set_sums(const float *target, uint64_t n, uint64_t max_depth, float *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PYASCORE_AUX_H */
