/*
 * pyascore_hip.h -- C ABI of the MI355X-native Ascore scorer (libpyascore_hip.so).
 *
 * This is the drop-in boundary for pyAscore's ptm_scoring hot path.  Every entry point
 * replaces a piece of the reference's Cython/C++ interface (citations are into the reference
 * tree, pyascore/ptm_scoring/):
 *
 *   pya_create / pya_destroy      PyAscore.__cinit__/__dealloc__      Ascore.pyx:64-79
 *   pya_add_neutral_loss          PyAscore.add_neutral_loss           Ascore.pyx:81-99
 *   pya_score_batch               PyAscore.score, batched             Ascore.pyx:103-152
 *                                 (= BinnedSpectra::consumeSpectra cpp/Spectra.cpp:43-68,
 *                                    ModifiedPeptide::consumePeptide/consumePeak
 *                                    cpp/ModifiedPeptide.cpp:105-142, Ascore::score
 *                                    cpp/Ascore.cpp:256-271)
 *   pya_results fields            best_score / ascores / alt_sites / best signature
 *                                                                     Ascore.pyx:232-288
 *   pya_get_pep_scores            PyAscore.pep_scores                 Ascore.pyx:241-252
 *   pya_get_pep_scores_range      the same for a range of PSMs of a retained batch (bulk export)
 *   pya_calculate_ambiguity       PyAscore.calculate_ambiguity        Ascore.pyx:208-230
 *   pya_format_peptide(s)         ModifiedPeptide::getPeptide         cpp/ModifiedPeptide.cpp:199-253
 *   pya_plan_*                    (new) device-resident variant of pya_score_batch for callers
 *                                 that keep spectra in HBM and own a HIP stream
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or framework types cross the boundary;
 *   - inputs are borrowed for the duration of the call (for pya_plan_run: until the stream
 *     has drained); outputs are caller-allocated; the library never frees caller memory;
 *   - every function returns PYA_OK (0) or a negative status; pya_last_error() gives the
 *     message and pya_error_index() the offending PSM (or -1);
 *   - a handle is single-owner (one handle <-> one device); different handles may be used
 *     from different threads;
 *   - there is NO CPU fallback: without a HIP device pya_create fails with PYA_ERR_HIP.
 *
 * Signature bit sets ("sig bits"): bit j is the j-th modifiable residue counted from the
 * N-terminus, 1 = modified.  Alternative-site masks: bit (p-1) = 1-based peptide position p.
 */
#ifndef PYASCORE_HIP_H
#define PYASCORE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYA_OK 0
#define PYA_ERR_ARG (-1)    /* invalid configuration / argument                          */
#define PYA_ERR_HIP (-2)    /* HIP runtime failure or no device                          */
#define PYA_ERR_PSM (-3)    /* a PSM is invalid (unknown residue, empty spectrum, ...)   */
#define PYA_ERR_LIMIT (-4)  /* a PSM exceeds a documented limit of this implementation   */
#define PYA_ERR_STATE (-5)  /* call sequence error (e.g. no retained batch)              */

/* documented limits (DESIGN.md "Limits").  The FAST_* values are what the fast kernels take; a PSM beyond one of
 * them (and inside the limits above it) is scored by the general kernel (csrc/general_psm.hip): same results, the
 * reference's algorithm at a fraction of the speed. */
#define PYA_MAX_PEPTIDE_LEN 511               /* (the general kernel; the reference takes any length) */
#define PYA_MAX_SITES 63                   /* (the reference's own limit: cpp/Ascore.cpp:91-94 keys a signature by a long) */
#define PYA_MAX_SIGNATURES (1 << 22)       /* C(n_sites, n_of_mod) */
#define PYA_MAX_FRAGMENTS_PER_TYPE 8192    /* (L - 1) x charges x neutral-loss sums */
#define PYA_FAST_PEPTIDE_LEN 64
#define PYA_FAST_SIGNATURES 15000
#define PYA_FAST_FRAGMENTS_PER_TYPE 2048
#define PYA_MAX_PEAKS 65535                /* peaks of one spectrum */
#define PYA_FAST_PEAKS 8192
#define PYA_MAX_FRAGMENT_TYPES 8
#define PYA_MAX_CHARGE 255                 /* max_fragment_charge (the fragments per ion type above bound it long before) */
#define PYA_MAX_NL_VALUES 8                /* distinct neutral-loss masses; more than 4: every PSM through the general kernel */
#define PYA_N_TOP 10                       /* the value everything is built for (the reference's command line passes 10) */
#define PYA_MAX_N_TOP 16                   /* 11..16: every PSM of the scorer goes through the general kernel            */

#define PYA_FLAG_KEEP 1u    /* retain per-signature records for pya_get_pep_scores /     */
                            /* pya_calculate_ambiguity                                    */
#define PYA_FLAG_TIMING 2u  /* record HIP events around every kernel of pya_plan_run     */
#define PYA_FLAG_SKIP_INVALID 4u /* a PSM that is invalid, exceeds a limit or is rejected by a  */
                            /* kernel does not fail the call: it gets best_score -1, n_sig -1, */
                            /* and its code in pya_last_batch_status(); the rest is scored      */

/* per-PSM codes of pya_last_batch_status */
#define PYA_PSM_OK 0
#define PYA_PSM_NO_WINDOWS 1       /* all peaks on one multiple of 100 m/z (reference: UB)      */
#define PYA_PSM_TOO_MANY_WINDOWS 2
#define PYA_PSM_TABLE_RANGE 3      /* trial count outside the score table                        */
#define PYA_PSM_TIED_OVERFLOW 4
#define PYA_PSM_INVALID 16         /* unknown residue, empty spectrum, bad charge, ...           */
#define PYA_PSM_OVER_LIMIT 17      /* beyond a documented limit (length, sites, C(n,k), peaks)   */

typedef struct pya_handle pya_handle;
typedef struct pya_plan pya_plan;

typedef struct pya_config {
    float bin_size;             /* m/z width of a peak window                            */
    uint32_t n_top;             /* peaks retained per window = depths scored: 10..16     */
    const char *mod_group;      /* e.g. "STY"; 'n' / 'c' allow the termini               */
    float mod_mass;
    float mz_error;             /* Da                                                    */
    const char *fragment_types; /* subset of "bycz Z", e.g. "by"                         */
    int32_t device;             /* HIP device ordinal                                    */
} pya_config;

/* host-side CSR description of a batch of PSMs (all arrays borrowed) */
typedef struct pya_batch {
    uint64_t n_psm;
    const int64_t *peak_off;    /* [n_psm+1] offsets into mz / intensity                 */
    const uint8_t *pep;         /* peptide letters of all PSMs, back to back             */
    const int64_t *pep_off;     /* [n_psm+1]                                             */
    const int32_t *n_of_mod;    /* [n_psm] unlocalised modifications                     */
    const int32_t *max_charge;  /* [n_psm] max fragment charge (>= 1)                    */
    const uint32_t *aux_pos;    /* fixed mods: 0 = n-term, else 1-based position; or NULL */
    const float *aux_mass;
    const int64_t *aux_off;     /* [n_psm+1] or NULL                                     */
} pya_batch;

/* caller-allocated structure-of-arrays results; host pointers for pya_score_batch,
 * device pointers for pya_plan_run */
typedef struct pya_results {
    uint32_t max_k;             /* row stride of ascores / alt_mask (>= max n_of_mod)    */
    float *best_score;          /* [n_psm]  PepScore of the best localisation, -1 if none */
    uint64_t *best_sig;         /* [n_psm]  sig bits of the best localisation            */
    int32_t *n_sig;             /* [n_psm]  number of localisations scored               */
    float *ascores;             /* [n_psm * max_k]  +inf where unambiguous               */
    uint64_t *alt_mask;         /* [n_psm * max_k] alternative sites of every modified site: bit p = residue p of the
                                 * peptide (0-based) for peptides of up to 64 residues; for longer ones (general
                                 * kernel) bit j = the j-th modifiable residue (pya_count_sites gives its position) */
} pya_results;

int pya_create(const pya_config *cfg, pya_handle **out);
void pya_destroy(pya_handle *h);
int pya_add_neutral_loss(pya_handle *h, const char *group, float mass);
/* PyAscore.score for ONE PSM with the lowest latency (Ascore.pyx:103-152; the reference's own command line
 * calls it once per PSM, pyascore/__main__.py:129-164): no plan, no copies -- the spectrum is staged in pinned
 * host memory the device reads directly, the PSM's scalars travel in the kernel's arguments, one wavefront runs
 * the whole path and writes the results straight back into pinned host memory.  Results as row 0 of `out`
 * (out->max_k <= 64).  flags: PYA_FLAG_KEEP retains the per-signature records at once; without it
 * pya_rescore_last_keep() retains them on demand (the properties only a few callers read).  Returns
 * PYA_ERR_STATE without an error message when the PSM needs the batch path (more than 8 fixed
 * modifications): call pya_score_batch then. */
int pya_score_one(pya_handle *h, const double *mz, const double *intensity, uint64_t n_peaks, const uint8_t *peptide,
                  uint64_t peptide_len, int32_t n_of_mod, int32_t max_fragment_charge, const uint32_t *aux_pos,
                  const float *aux_mass, uint64_t n_aux, uint32_t flags, const pya_results *out);
int pya_rescore_last_keep(pya_handle *h);

/* The environment reaches a handle through four variables, read once in pya_create: PYA_WORKSPACE_MB and
 * PYA_CHUNK_MB (sizes of pya_score_batch's device budget and chunks), PYA_HOST_TIMING and PYA_STAMPS (diagnostics that
 * change no result and no route).  This re-reads them and puts every debug switch (include/pyascore_debug.h) back to
 * its production default.  No reference counterpart. */
int pya_reload_env(pya_handle *h);
/* diagnostics: average microseconds per pya_score_one call since the last call of this function, by stage (checks
 * and tables, spectrum into the pinned block, launch, wait for the kernel, results out); us[5] = calls averaged;
 * us[6..9] = inside the kernel by its own clock (scalars into place, binning, scoring, the rest), us[10] = the
 * kernel's shader clock cycles */
int pya_one_times(pya_handle *h, double us[12]);

const char *pya_last_error(const pya_handle *h);
int64_t pya_error_index(const pya_handle *h);

/* host buffers in, host buffers out: H2D copy + kernels + D2H copy, synchronous (big batches are
 * chunked and pipelined, see pya_set_workspace_budget; a PYA_FLAG_KEEP batch is always one plan) */
int pya_score_batch(pya_handle *h, const pya_batch *batch, const double *mz,
                    const double *intensity, uint32_t flags, const pya_results *out);

/* Device memory one pya_score_batch call may hold at a time (upload ring + workspace; default 6 GiB,
 * or PYA_WORKSPACE_MB).  Calls that need more -- and every call with more than 32 MB of spectra -- are
 * cut into chunks of consecutive PSMs and pipelined: the upload of chunk c + 1 runs under the kernels
 * and the result copy of chunk c.  Results do not depend on the cut. */
int pya_set_workspace_budget(pya_handle *h, uint64_t bytes);
/* the budget in force (the value set, PYA_WORKSPACE_MB, or the default of 6 GiB) */
uint64_t pya_get_workspace_budget(const pya_handle *h);

/* per-PSM status codes (PYA_PSM_*) of the last pya_score_batch call on this handle; n must equal
 * that batch's n_psm.  All zeros unless PYA_FLAG_SKIP_INVALID let PSMs be set aside. */
int pya_last_batch_status(pya_handle *h, int32_t *status, uint64_t n);

/* device-resident path: plan once (host pre-pass, tables, workspace), run many times */
int pya_plan_create(pya_handle *h, const pya_batch *batch, uint32_t flags, pya_plan **out);
int pya_plan_run(pya_plan *plan, const double *d_mz, const double *d_intensity,
                 void *hip_stream, const pya_results *d_out);
/* ms per kernel family of the last pya_plan_run (PYA_FLAG_TIMING): bin_spectra, score_signatures,
 * score_localize (the fused kernel, with the localize launch for what it hands over), localize;
 * synchronises */
int pya_plan_timings(pya_plan *plan, float ms[4]);
/* the same, summed over the runs since the last call of this function (the events live in a ring of 128 runs:
 * of more runs than that only the latest 128 count); *n_runs = how many; synchronises with the latest run only,
 * so a caller can enqueue run after run without waiting in between */
int pya_plan_timings_sum(pya_plan *plan, double ms[4], uint32_t *n_runs);
/* The multi-GPU path's gather record (north_star: "a single RCCL gather at the end"; no reference counterpart -- the
 * reference scores on one core and has no collective): packs the device results of `n_psm` PSMs into fixed-size
 * records of 4 + 3 k int32 words (best_score bits, n_sig, best_sig lo / hi, k Ascore bit patterns, k alternative-site
 * masks lo / hi; k >= d_res->max_k, the job-wide widest row, columns beyond the results' own are zero) at d_out --
 * e.g. a rank's slice of its send buffer -- with ONE kernel on `hip_stream`. */
int pya_pack_records(pya_handle *h, const pya_results *d_res, uint64_t n_psm, uint32_t k, int32_t *d_out, void *hip_stream);
/* waits for the stream of the last run and reports the first PSM the kernels rejected */
int pya_plan_check(pya_plan *plan);
uint64_t pya_plan_workspace_bytes(const pya_plan *plan);
uint64_t pya_plan_total_signatures(const pya_plan *plan);
void pya_plan_destroy(pya_plan *plan);

/* all localisations of PSM `psm` of the last pya_score_batch(..., PYA_FLAG_KEEP, ...), in the
 * reference's sorted order; arrays sized for `cap` records; returns the record count in *n */
int pya_get_pep_scores(pya_handle *h, uint64_t psm, uint64_t cap, uint64_t *n, uint64_t *sig_bits,
                       int32_t *counts /* cap x n_top */, float *scores /* cap x n_top */,
                       float *weighted_score, int32_t *total_fragments);
/* the same for PSMs [psm_begin, psm_end) in one call (three device copies for the whole range):
 * rec_off[psm_end - psm_begin + 1] receives the CSR offsets of the PSMs' records; with cap == 0 only
 * rec_off is filled (size query), otherwise the arrays must hold rec_off[last] records.
 * (pep_scores of a whole batch, Ascore.pyx:241-252 at scale) */
int pya_get_pep_scores_range(pya_handle *h, uint64_t psm_begin, uint64_t psm_end, uint64_t cap,
                             int64_t *rec_off, uint64_t *sig_bits, int32_t *counts /* cap x n_top */,
                             float *scores /* cap x n_top */, float *weighted_score,
                             int32_t *total_fragments);
/* PyAscore.calculate_ambiguity (Ascore.pyx:208-230 -> Ascore::calculateAmbiguity, cpp/Ascore.cpp:157-210) for PSM `psm`
 * of the retained batch: the caller's two score containers as (signature bits over the modifiable residues, n_top depth
 * scores, weighted score).  Any PSM the library scores: peptides of up to 64 residues with ten depths and a spectrum of
 * up to PYA_FAST_PEAKS peaks on the fast kernel, everything else (long peptides, n_top 11..16, big spectra) on the
 * general kernel's Ascore code. */
int pya_calculate_ambiguity(pya_handle *h, uint64_t psm, uint64_t ref_bits,
                            const float *ref_scores, float ref_weighted, uint64_t other_bits,
                            const float *other_scores, float other_weighted, float *out);

/* host-only helpers */
int pya_format_peptide(const pya_handle *h, const uint8_t *pep, uint64_t pep_len, int32_t n_of_mod,
                       const uint32_t *aux_pos, const float *aux_mass, uint64_t n_aux,
                       uint64_t sig_bits, int32_t sig_len, char *buf, uint64_t cap);
/* pya_format_peptide for many records in ONE call: record r is the localisation sig_bits[r] of PSM
 * rec_psm[r] of `batch` (rec_psm NULL: record r belongs to PSM r -- the best sequences of a scored
 * batch; with rec_psm -- the sequence column of a bulk pep_scores export).  rec_valid (optional): a
 * record with rec_valid[r] <= 0 (n_sig of a PSM without localisation, a set-aside PSM) gets the empty
 * string.  Strings come back as CSR bytes without terminators: str_off[n_rec + 1], buf[cap]; with
 * cap == 0 only str_off is filled (size query).  (ModifiedPeptide.cpp:199-253 at scale) */
int pya_format_peptides(const pya_handle *h, const pya_batch *batch, uint64_t n_rec, const int64_t *rec_psm,
                        const uint64_t *sig_bits, const int32_t *rec_valid, int64_t *str_off, char *buf,
                        uint64_t cap);
int pya_count_sites(const pya_handle *h, const uint8_t *pep, uint64_t pep_len, int32_t *n_sites,
                    uint16_t *site_pos /* >= PYA_MAX_PEPTIDE_LEN, 0-based residue of each site */);

/* test hook: runs the on-device emulation of the reference's std::sort (descending, keyed by
 * weighted score) on arbitrary keys and returns the permutation */
int pya_debug_sort(pya_handle *h, const float *keys, uint32_t n, uint32_t *perm);

const char *pya_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PYASCORE_HIP_H */
