/* common.h -- structures shared by the host library and the gfx950 kernels. */
#ifndef PYA_COMMON_H
#define PYA_COMMON_H

#include <stdint.h>

#define PYA_WAVE 64
#define PYA_NTOP 10                /* peaks retained per window / depths the fast kernels are built for              */
#define PYA_NTOP_MAX 16            /* ... the binning and the general kernel take (DevConfig.n_top)                   */
#define PYA_NO_MATCH 15            /* rank value meaning "no retained peak in the window" */
#define PYA_TABLE_PAD 4            /* +inf sentinels after the last retained peak of a staged table */
#define PYA_MAX_L 64
#define PYA_MAX_TYPES 8
#define PYA_MAX_NL 8               /* distinct neutral-loss masses (the general kernel)    */
#define PYA_FAST_NL 4              /* ... the fast kernels take (2 bits each in an 8-bit loss state, <= 16 distinct sums) */
#define PYA_MAX_UNIQ_WIDE 48       /* distinct sums of <= 2 of eight masses: 1 + 8 + 36    */
#define PYA_MAX_NL_CANDS 44        /* singles + pairs                                      */
#define PYA_MAX_UNIQ 16            /* distinct sums of <= 2 neutral losses (incl. 0)       */
#define PYA_MAX_LIST 2048          /* fragments of one signature and one ion type          */
#define PYA_MAX_LUT_N 4096         /* largest trial count the score table covers           */
#define PYA_MAX_PUSHED 128         /* single-move competitors of one PSM: k * (n_sites - k) <= 126 */
                                   /* for every shape with C(n,k) <= PYA_MAX_SIGNATURES             */

/* per-PSM status written by the kernels */
#define PYA_ST_OK 0
#define PYA_ST_NO_BINS 1           /* min == max at a multiple of 100: reference is UB      */
#define PYA_ST_TOO_MANY_BINS 2
#define PYA_ST_LUT_RANGE 3         /* trial count outside the uploaded score table          */
#define PYA_ST_PUSHED_OVERFLOW 4
#define PYA_ST_ROUTE_CAPS 5        /* a PSM beyond the caps its launch was sized for (a routing bug: never expected)      */
#define PYA_ST_INVALID 16          /* set aside by the host pre-pass (PYA_FLAG_SKIP_INVALID): invalid PSM    */
#define PYA_ST_OVER_LIMIT 17       /* ... or one that exceeds a documented limit of this implementation     */

/* Scorer configuration as the kernels see it (one copy in device memory per handle).
 * Residue tables are indexed by (letter - 'A') & 31.                                      */
struct DevConfig {
    float bin_size;
    float mod_mass;
    float mz_error;
    int32_t n_types;
    int32_t n_fwd;                  /* number of b/c entries in types[] (they come first)   */
    uint8_t types[PYA_MAX_TYPES];   /* forward types first, then backward; order is free    */
    uint8_t first_forward;          /* direction of fragment_types[0]                       */
    uint8_t allow_n, allow_c;       /* 'n' / 'c' in mod_group                               */
    uint8_t n_nl;                   /* distinct NL masses D                                 */
    float res_mass[32];             /* Types.h:7-30, 0 = not a residue                      */
    uint8_t res_modifiable[32];     /* letter in mod_group                                  */
    uint8_t nl_upper[32];           /* NL class (1..D, 0 none) of the unmodified residue    */
    uint8_t nl_lower[32];           /* NL class of the modified / aux-modified residue      */
    int32_t n_uniq;                 /* distinct values among {0} U {t_i} U {t_i + t_j}      */
    float uniq[PYA_MAX_UNIQ];       /* uniq[0] = 0                                          */
    uint16_t present[256];          /* NL stack state (2 bits per class, saturating at 2)   */
                                    /*   -> bit set of uniq[] values that exist             */
    float weights[PYA_NTOP];        /* Ascore.cpp:16-18                                     */
    int32_t n_top;                  /* peaks retained per window = depths scored (Ascore.pyx:64-67); 10 unless the general kernel runs everything */
    /* the same neutral-loss sums as the general kernel reads them (any n_nl <= PYA_MAX_NL; uniq[] / present[] above exist
     * for n_nl <= PYA_FAST_NL only): candidate i = the loss of class cand_a (cand_b == 255) or the sum of classes cand_a
     * and cand_b (equal: the class twice), present when the classes occurred that often; it is sum number cand_u */
    int32_t n_cand;
    float uniq_w[PYA_MAX_UNIQ_WIDE];
    uint8_t cand_a[PYA_MAX_NL_CANDS], cand_b[PYA_MAX_NL_CANDS], cand_u[PYA_MAX_NL_CANDS];
};

/* device pointers + scalars of one launch family; passed by value as kernel argument */
struct BatchDev {
    /* inputs */
    const double *mz;
    const double *inten;
    const int64_t *peak_off;
    const uint8_t *pep;
    const int64_t *pep_off;
    const int32_t *n_of_mod;
    const int32_t *max_charge;
    const uint32_t *aux_pos;
    const float *aux_mass;
    const int64_t *aux_off;         /* never NULL on device (all zeros if no aux mods)      */
    /* host pre-pass */
    const uint8_t *n_sites;         /* [n_psm]                                              */
    const uint32_t *n_sig;          /* [n_psm] C(n_sites, k) or 0                           */
    const uint32_t *order_off;      /* [n_psm] offset of the PSM's shape in order_tab       */
    const int64_t *sig_off;         /* [n_psm+1] offsets into ws / rec                      */
    const uint64_t *order_tab;      /* pre-sort signature order per shape (sig bits)        */
    const uint32_t *inv_tab;        /* same offsets: combination rank of a signature -> its index in order_tab */
    const uint32_t *binom;          /* [64][64] C(p, t), saturated: the combination rank is sum C(p_t, t)       */
    const uint64_t *desc;           /* [n_psm][PYA_DESC_WORDS] the offsets and counts above, packed (one  */
                                    /* cache line per PSM): ret_off, pep_off, sig_off, aux_off,           */
                                    /* L | n_aux << 16 | n_of_mod << 32 | n_sites << 48 | max_charge << 56, n_sig | order_off << 32 */
    const DevConfig *cfg;
    const float *lut;               /* score table                                          */
    const uint32_t *lut_off;        /* [lut_n_max+1] row offsets                            */
    uint32_t lut_n_max;
    /* workspace */
    struct PeakEntry *ret;          /* retained peaks (float m/z, rank), m/z ascending, at ret_off[psm]: 8-byte  */
                                    /* entries from an even (16-byte aligned) offset, so a lane moves two at a time */
    const int64_t *ret_off;         /* [n_psm] even; room for the PSM's raw peak count rounded up to even           */
    uint32_t *ret_n;                /* [n_psm]                                              */
    uint16_t *grid;                 /* [n_psm][PYA_GRID_CELLS] m/z grid over the retained peaks (score_signatures) */
    uint32_t *zero_next;            /* [3] or NULL: the hand-over counts of the plan's NEXT run, zeroed by this run's binning kernel */
    uint32_t *redo_count;           /* spectra bin_spectra hands to its exact variant (peaks out of */
    uint32_t *redo_ids;             /* [n_psm] m/z order, or equal intensities inside a window)     */
    uint32_t *redo3_count;          /* PSMs the lean localize instantiation hands to the general one */
    uint32_t *redo3_ids;            /* [n_psm]                                                       */
    uint32_t *redo3b_count;         /* ... and what the lean instantiation's second (sorting) pass declines */
    uint32_t *redo3b_ids;           /* [n_psm]                                                       */
    uint32_t *redo4_count;          /* PSMs the fused score + localize kernel hands to the general   */
    uint32_t *redo4_ids;            /* [n_psm] localize instantiation                                */
    float *ws;                      /* weighted score per signature, pre-sort order         */
    uint32_t *ws_top;               /* [n_psm][4] score_signatures' summary of ws: largest value (bits), how many */
                                    /* signatures have it (0 = not known), the first of them; localize starts here */
    uint32_t *rec;                  /* optional per-signature records: 6 words each         */
    uint32_t *sorted_idx;           /* optional sorted permutation, at sig_off              */
    int32_t *status;                /* [n_psm]                                              */
    /* outputs */
    float *best_score;
    uint64_t *best_sig;
    int32_t *n_sig_out;
    float *ascores;
    uint64_t *alt_mask;
    uint32_t max_k;
    uint32_t keep;                  /* write rec / sorted_idx                               */
    /* PYA_DEBUG bits.  Ablation, results become meaningless: 1 no site-determining ions, 4 no prefix
     * tables, 8 no sort emulation, 16 no competitors, 32 no window ranking, 64 no compaction.
     * Route selection, results stay exact (used by the tests): 128 every spectrum through
     * pya_bin_exact_kernel, 512 the lean localize instantiation declines every PSM, 1024 the
     * std::sort emulation runs even for a unique best PepScore, 2048 the fused kernel replays every
     * (competitor, direction) task with the serial walk (and pairs ions over whole lists, not spans),
     * 4096 the list-based general localize instantiation replays a task with a doubly partnered ion serially
     * instead of walking its clusters in parallel, 8192 the hash route of the general localize declines every
     * PSM (hand-over list, list-based kernel), 16384 it sends every in-span ion through the exact run walk,
     * 256 score_signatures walks every signature under general settings (no shared tree nodes), 0x8000 no
     * count-node table (walk_core.hip.h: every walker looks every fragment up itself), 0x40000000 every node of that
     * table marked (the table is read, then every walker looks up itself), 0x20000000 the hash route sends an ion with a lone
     * neighbour through the exact run walk too (no closed form for {ion, winner's ion outside the span, its twin}), 0x10000000
     * score_big leaves no candidate records (the PepScores of all site assignments go to the workspace, the finishing kernel
     * ranks, gathers and recounts: the r05 route).  Ablation: 2 the hash route looks no surviving ion up.
     * Bits 16..31: truncation point of the diagnostic build (device_common.hip.h, STAMP_T). */
    uint32_t debug;
    unsigned long long *stamps;     /* per-phase cycle sums (diagnostic build -DPYA_STAMPS)  */
};

/* one retained peak: float m/z and its rank inside its window, 8 bytes so both come with one 8-byte read
 * (two entries with one 16-byte read) from LDS and from the workspace alike */
struct PeakEntry {
    float mz;
    uint32_t rank;
};

/* one tied best competitor as the scan leaves it in LDS */
struct PushedEntry {
    uint64_t bits;   /* signature */
    float ws;        /* its PepScore */
    uint32_t idx;    /* its pre-sort index */
};

/* The scalars of ONE PSM, handed to pya_one_kernel in its kernel arguments (the dispatch packet is the
 * fastest way to get a few bytes to the device: no copy, no second round trip); the wavefront writes them
 * where the per-PSM bodies expect the batch arrays. */
#define PYA_ONE_MAX_AUX 8
struct OneMeta {
    uint64_t pep[PYA_MAX_L / 8];    /* the peptide's letters */
    uint64_t desc[6];               /* the packed descriptor (BatchDev.desc) */
    uint32_t n_peaks, L, n_aux, n_sig, order_off, seq;
    int32_t n_of_mod, max_charge;
    uint32_t n_sites, pad;
    uint32_t aux_pos[PYA_ONE_MAX_AUX];
    float aux_mass[PYA_ONE_MAX_AUX];
};

/* LDS bytes of the batched-localisation work area (localize_core.hip.h: LocLds) */
#define PYA_GRID_CELLS 256         /* cells of the m/z grid that accelerates the peak lookup    */
#define PYA_LOC_SB_MAX 8           /* signatures worked on together: the winner + 7 competitors */
static inline unsigned long pya_loc_lds_bytes(unsigned pos_cap, unsigned pool_cap, unsigned sb) {
    return 64ul * 4 * 2 + 64 + sb * 8ul + (unsigned long)sb * 2 * pos_cap * 8 + sb * 2 * 4ul +
           sb * 10 * 4ul + sb * 4 * 3ul + sb * 2 * 4 * 2ul + (unsigned long)pool_cap * 6 + 128 * 8 + 64 +
           (32 + 33) * 4ul;                                       /* span tables of the lean pairing */
}

/* per-signature record: 10 cumulative counts as u16 + total fragments */
#define PYA_REC_WORDS 6

/* packed per-PSM descriptor (BatchDev.desc) */
#define PYA_DESC_WORDS 6

#endif
