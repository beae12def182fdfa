/* host_internal.h -- host side of libpyascore_hip.so: the C ABI of include/pyascore_hip.h.
 *
 * Host work is limited to what is not data-parallel arithmetic over spectra:
 *   - validating PSMs and counting modifiable residues (one pass over the peptide letters),
 *   - per-shape signature order tables (the iteration order of the reference's
 *     std::unordered_map<long,...>, cpp/Ascore.cpp:54,114-120 -- reproduced with the same
 *     libstdc++ container, it depends only on (n_sites, n_mods, direction)),
 *   - the binomial score table (score_table.cpp),
 *   - workspace sizing, bucketing PSMs by C(n,k) so each launch gets the LDS it needs,
 *   - kernel launches and copies.
 * There is no CPU scoring path here: without a HIP device every entry point fails.
 *
 * This header holds what the host files share (handle, plan, buckets, device buffers, the kernels' launchers);
 * the code is in host_abi.cpp (handles, settings, strings, retained records), host_tables.cpp (DevConfig, score table,
 * order tables, environment switches), host_plan.cpp (plans: pre-pass, arena, launches), host_batch.cpp
 * (pya_score_batch: chunking and pipelining) and host_one.cpp (pya_score_one).
 */
#ifndef PYA_HOST_INTERNAL_H
#define PYA_HOST_INTERNAL_H
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pyascore_hip.h"
#include "common.h"

void pya_score_table_extend(float mz_error, uint32_t n_top, uint32_t n_to, std::vector<float> &lut,
                            std::vector<uint32_t> &off);
extern "C" {
size_t pya_bin_lds_bytes(uint32_t cap);
size_t pya_score_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact);
size_t pya_localize_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb);
int pya_launch_bin(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, hipStream_t stream);
int pya_launch_bin_select(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t scap, hipStream_t stream);
size_t pya_bin_select_lds_bytes(uint32_t scap);
int pya_launch_bin_exact(const BatchDev *b, uint32_t n_total, uint32_t cap, hipStream_t stream);
int pya_launch_score(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t prefix,
                     uint32_t with_nl, uint32_t compact, uint32_t node_cap, uint32_t node_cols, uint32_t node_words,
                     uint32_t res_cap, uint32_t nl_cap, hipStream_t stream);
size_t pya_score_node_lds_bytes(uint32_t cap, uint32_t with_nl, uint32_t node_cap, uint32_t node_cols, uint32_t node_words,
                                uint32_t res_cap, uint32_t nl_cap);
int pya_launch_localize(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t push_cap,
                        uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp,
                        uint32_t plain, uint32_t sort_room, hipStream_t stream);
size_t pya_tiny_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact, uint32_t push_cap,
                          uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb);
int pya_launch_tiny(const BatchDev *b, uint32_t n_psm, uint32_t cap, uint32_t prefix, uint32_t with_nl,
                    uint32_t compact, uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap,
                    uint32_t sb, uint32_t gtp, hipStream_t stream);
size_t pya_one_lds_bytes(uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact, uint32_t push_cap, uint32_t n_cap,
                         uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t use_fused, uint32_t f_n_cap, uint32_t f_stride,
                         uint32_t f_ent_cap, uint32_t f_push_cap, uint32_t multi_z);
int pya_launch_one(const BatchDev *b, const OneMeta *m, uint32_t cap, uint32_t prefix, uint32_t with_nl, uint32_t compact,
                   uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp,
                   uint32_t use_fused, uint32_t f_n_cap, uint32_t f_stride, uint32_t f_ent_cap, uint32_t f_push_cap,
                   uint32_t multi_z, int32_t *host_status, uint32_t *host_flag, hipStream_t stream);
int pya_launch_ambiguity(const BatchDev *b, uint32_t psm, uint32_t peak_cap, uint32_t list_cap,
                         uint64_t ref_bits, uint64_t oth_bits, const float *d_scores, float ref_ws,
                         float oth_ws, float *d_out, hipStream_t stream);
int pya_launch_debug_sort(const float *d_keys, uint32_t n, uint32_t *d_perm, hipStream_t stream);
int pya_launch_pack_records(const float *best_score, const int32_t *n_sig, const uint64_t *best_sig, const float *ascores,
                            const uint64_t *alt_mask, uint32_t k, uint32_t res_k, uint64_t n_psm, int32_t *out, hipStream_t stream);
int pya_launch_debug_wave_ops(const int32_t *d_in, int32_t *d_out, hipStream_t stream);
size_t pya_score_big_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc);
int pya_launch_score_big(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap, uint32_t kc,
                         uint32_t inline_on, hipStream_t stream);
size_t pya_localize_recount_lds_bytes(uint32_t cap, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb);
size_t pya_localize_hash_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc, uint32_t hs,
                                   uint32_t pp, uint32_t max_k, uint32_t n_nl);
int pya_launch_localize_hash(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t push_cap, uint32_t n_cap,
                             uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp, uint32_t vc, uint32_t hs, uint32_t pp,
                             uint32_t n_nl, hipStream_t stream);
int pya_launch_localize_recount(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t push_cap,
                                uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp, uint32_t *d_redo_count, uint32_t *d_redo_ids, hipStream_t stream);
int pya_launch_score_big_list(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max, uint32_t cap,
                              uint32_t pos_cap, uint32_t kc, hipStream_t stream);
uint32_t pya_big_inline_max(void);
size_t pya_fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap,
                           uint32_t both, uint32_t multi_z);
int pya_launch_fused(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap,
                     uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap, uint32_t both,
                     uint32_t multi_z, uint32_t *d_redo_count, uint32_t *d_redo_ids, hipStream_t stream);
size_t pya_score_cnt_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap, uint32_t n_cap);
int pya_launch_score_cnt(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap, uint32_t kc, uint32_t k_cap,
                         uint32_t n_cap, hipStream_t stream);
size_t pya_score_cntg_lds_bytes(uint32_t cap, uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap, uint32_t nl_cap);
int pya_launch_score_cntg(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap, uint32_t k_cap, uint32_t n_cap,
                          uint32_t nl_cap, hipStream_t stream);
size_t pya_bin_global_scratch_bytes(uint32_t cap);
int pya_launch_bin_global(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch, uint64_t stride,
                          uint32_t cap, hipStream_t stream);
size_t pya_general_lds_bytes(uint32_t l_cap, uint32_t list_cap);
size_t pya_general_scratch_bytes(uint32_t n_cap, uint32_t push_cap);
int pya_launch_general(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch, const uint64_t *d_scratch_off,
                       uint32_t l_cap, uint32_t list_cap, hipStream_t stream);
int pya_launch_general_ambiguity(const BatchDev *b, uint32_t psm, uint32_t l_cap, uint32_t list_cap, uint64_t ref_bits,
                                 uint64_t oth_bits, const float *d_scores, uint32_t n_scores, float ref_ws, float oth_ws,
                                 float *d_out, hipStream_t stream);
int pya_launch_localize_redo(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max,
                             uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                             uint32_t gtp, hipStream_t stream);
}


/* (shared by the host_*.cpp files; internal linkage where it is data, inline where it is code) */

const size_t kMaxLds = 160 * 1024;
const uint32_t kBucketLimits[] = {64, 512, 4096, PYA_FAST_SIGNATURES};
const int kNumBuckets = 4;
const uint64_t kTinyBatch = 64;         /* up to this many PSMs go through the fused single-launch kernel */
const size_t kStageLimit = 1u << 20;   /* batches whose transfers are smaller than this go through one staged copy */

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        n = 0;
        owned = true;
    }
    /* points into somebody else's allocation (the plan's arena) */
    void adopt(void *ptr, size_t count) {
        release();
        p = (T *)ptr;
        n = count;
        owned = false;
    }
    hipError_t fill(const T *src, hipStream_t s = nullptr) {
        if (n == 0 || !src) return hipSuccess;
        return hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, s);
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    hipError_t upload(const T *src, size_t count, hipStream_t s = nullptr) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
    size_t bytes() const { return n * sizeof(T); }
    /* moves the allocation of `other` here if it is big enough; returns whether it did */
    bool take_if_fits(DevBuf &other, size_t count) {
        if (!other.p || !other.owned || other.n < count) return false;
        release();
        p = other.p;
        n = other.n;
        other.p = nullptr;
        other.n = 0;
        return true;
    }
    void give_to(DevBuf &other) {
        if (!p || !owned) return;
        if (other.p && other.n >= n) return;          /* keep the larger one */
        other.release();
        other.p = p;
        other.n = n;
        p = nullptr;
        n = 0;
    }
};

inline uint64_t binom(uint32_t n, uint32_t k) {
    if (k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned __int128 r = 1;
    for (uint32_t i = 1; i <= k; i++) {
        r = r * (n - k + i) / i;
        if (r > (unsigned __int128)1 << 62) return ~0ull;
    }
    return (uint64_t)r;
}

inline bool is_forward(char t) { return t == 'b' || t == 'c'; }
inline bool is_backward(char t) { return t == 'y' || t == 'z' || t == 'Z'; }

inline float std_residue_mass(char c) {                       /* Types.h:7-30 */
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


/* Debug switches (route selection for the tests, diagnostics, A/B experiments): per handle, set through
 * pya_set_debug (include/pyascore_debug.h) -- NOT from the environment, which reaches the library through four
 * variables only (host_tables.cpp:read_env: PYA_WORKSPACE_MB, PYA_CHUNK_MB, PYA_HOST_TIMING, PYA_STAMPS; read in
 * pya_create, re-read by pya_reload_env).  Defaults are the production behaviour. */
struct Knobs {
    bool no_plain = false, no_fused = false, no_big = false, no_tiny = false, no_prefix = false, no_chunks = false;
    bool no_upload_thread = false, one_peak_class = false, peak_classes = false, one_lds_class = false;
    bool host_timing = false, stamps = false, sort_room = false, no_big_inline = false;
    bool no_loc_hash = false, no_nodes = false, no_cnt = false;
    bool no_fork = false;                       /* r06: the fused family on the plan's stream behind the scoring kernels instead of beside them */
    bool slow_null_stream = false;              /* host_one.cpp: widens the window of a (fixed) workspace race for its regression test */
    uint32_t debug = 0;
    int64_t plain_min = 512, big_min_n = 1024, tiny_max = 64;
    int64_t bin_select_min = 640;               /* peak classes above this many peaks are binned by selection (bin_select.hip.h); tests: 0 = all, huge = none */
    int64_t bin_select_scap = 768;              /* survivor slots per spectrum there */
    uint32_t sort_room_max = 1024;
    int sb = -1, gtp = -1, hash_pp = -1;        /* < 0: the built-in rule */
    int node_cap = -1;                          /* >= 0: room for that many shared nodes per direction (tests: small values force the walkers) */
    double chunk_mb = 0.;                       /* 0: the default chunk size */
    int64_t workspace_mb = 0;                   /* 0: the default budget */
};

void read_knobs(Knobs &k);
bool set_knob(Knobs &k, const char *key, const char *value);

struct pya_handle {
    Knobs kn;
    int device = 0;
    hipStream_t side_stream = nullptr;        /* r06: where the plans' forked fused families run (pya_plan::fork) */
    float bin_size = 100.f, mod_mass = 0.f, mz_error = 0.5f;
    uint32_t n_top = PYA_NTOP;                /* 10: the fast kernels; 11..16: every PSM through the general kernel */
    uint32_t rec_words() const { return (n_top + 1u) / 2u + 1u; }   /* count record: n_top 16-bit counts + the fragment total */
    std::string mod_group, fragment_types;
    std::map<char, float> nl;                 /* letter -> neutral loss (ModifiedPeptide.h:19) */
    DevConfig cfg;
    bool cfg_dirty = true;
    DevBuf<DevConfig> d_cfg;

    std::vector<float> lut;
    std::vector<uint32_t> lut_off;
    uint32_t lut_uploaded_n = 0;              /* rows [0, lut_uploaded_n) are on the device */
    DevBuf<float> d_lut;
    DevBuf<uint32_t> d_lut_off;

    std::map<uint32_t, uint32_t> shape_off;   /* (n << 8 | k) -> offset into order_tab */
    std::map<uint32_t, uint32_t> shape_cols;  /* ... -> histogram columns its shared-node route needs (shapes of <= 64 signatures) */
    std::vector<uint64_t> order_tab;
    std::vector<uint32_t> inv_tab;            /* same offsets: combination rank -> index in order_tab */
    size_t order_uploaded = 0;
    DevBuf<uint64_t> d_order;
    DevBuf<uint32_t> d_inv, d_binom;

    /* device allocations recycled between pya_score_batch calls (hipMalloc/hipFree of a few
     * hundred MB cost milliseconds) */
    DevBuf<unsigned char> spare_arena, spare_arena2;   /* two: chunked calls keep two plans alive */
    void *pinned_stage[2] = {nullptr, nullptr};        /* chunked calls: results of chunk c land in slot c % 2 */
    size_t pinned_bytes[2] = {0, 0};
    DevBuf<double> io_buf;                     /* spectra of big pya_score_batch calls (uploaded by a helper thread) */
    DevBuf<double> io_ring[2];                 /* chunked calls: spectra of chunk c in slot c % 2 */
    hipStream_t copy_stream = nullptr, run_stream = nullptr;   /* chunked calls: uploads / kernels + results */
    size_t ws_budget = 0;                      /* device bytes one pya_score_batch call may hold (0 = default) */
    std::vector<unsigned char> stage;          /* host staging of small batches: one copy each way */

    /* pya_score_one: persistent pinned (device-mapped, coherent) host block + one PSM's device workspace */
    struct One {
        unsigned char *host = nullptr, *host_dev = nullptr;    /* the same block as the host / the device sees it */
        DevBuf<unsigned char> ws, probe;          /* (probe: the PYA_SLOW_NULL_STREAM test switch's 256 MB) */
        uint32_t sig_cap = 0;                      /* signatures the workspace has room for */
        uint32_t seq = 0;
        hipStream_t stream = nullptr;
        BatchDev dev;
        OneMeta meta;                              /* of the last call (pya_rescore_last_keep replays it) */
        bool have_last = false, last_keep = false;
        uint32_t last_max_k = 1;
        pya_plan *view = nullptr;                  /* what pya_get_pep_scores / pya_calculate_ambiguity read */
        double t_sum[5] = {0, 0, 0, 0, 0};         /* seconds in checks + tables, copy in, launch, wait, copy out (pya_one_times) */
        double t_dev[4] = {0, 0, 0, 0};            /* seconds inside the kernel: scalars, binning, scoring, rest */
        double t_cycles = 0;                       /* its shader clock cycles */
        uint64_t t_calls = 0;
    } one;

    std::string err;
    int64_t err_index = -1;
    std::vector<int32_t> last_status;         /* per-PSM codes of the last pya_score_batch */
    pya_plan *kept = nullptr;                 /* plan of the last PYA_FLAG_KEEP batch */
    /* settings only the general kernel takes: every PSM of the scorer goes there (cfg is rebuilt by every setter) */
    bool all_general() const { return n_top != PYA_NTOP || cfg.n_nl > PYA_FAST_NL; }

    int fail(int code, int64_t index, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        err_index = index;
        return code;
    }
    int hip_fail(hipError_t e, const char *what) {
        return fail(PYA_ERR_HIP, -1, "HIP error in %s: %s", what, hipGetErrorString(e));
    }
    uint64_t binom_cache[64][64] = {{0}};     /* C(n,k), 0 = not computed yet (C >= 1 always) */
    uint32_t shape_cache[64][64];             /* offset of the shape's order table, ~0 = unknown */
    uint8_t in_group[256] = {0};              /* letter is in mod_group */
    uint8_t is_residue[256] = {0};            /* letter has a mass in Types.h */
    uint8_t letter_cls[256] = {0};            /* bit 0: residue of Types.h, bit 1: in mod_group */
    bool allow_n = false, allow_c = false;
    /* validity of the letters and the number of modifiable residues of one peptide (L >= 1) */
    bool scan_peptide(const uint8_t *s, int64_t L, uint32_t *n_sites) const {
        uint32_t all = 1, ns = 0;
        for (int64_t j = 0; j < L; j++) {
            const uint32_t c = letter_cls[s[j]];
            all &= c;
            ns += c >> 1;
        }
        if (allow_n && !(letter_cls[s[0]] >> 1)) ns++;
        if (allow_c && !(letter_cls[s[L - 1]] >> 1) && !(L == 1 && allow_n)) ns++;
        *n_sites = ns;
        return all & 1u;
    }
    void build_letter_tables() {
        std::memset(in_group, 0, sizeof in_group);
        std::memset(is_residue, 0, sizeof is_residue);
        for (unsigned char c : mod_group) in_group[c] = 1;
        for (int c = 'A'; c <= 'Z'; c++) is_residue[c] = std_residue_mass((char)c) != 0.f;
        for (int c = 0; c < 256; c++) letter_cls[c] = (uint8_t)((is_residue[c] ? 1 : 0) | (in_group[c] ? 2 : 0));
        allow_n = mod_group.find('n') != std::string::npos;
        allow_c = mod_group.find('c') != std::string::npos;
    }
    bool letter_modifiable(char c, size_t i, size_t L) const {
        return in_group[(unsigned char)c] || (i == 0 && allow_n) || (i + 1 == L && allow_c);
    }
};

#define HIPCHK(h, call)                                        \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return (h)->hip_fail(e_, #call); \
    } while (0)

struct Bucket {
    std::vector<uint32_t> ids;          /* every PSM of the bucket (score_signatures launch), plain ones first */
    std::vector<uint32_t> general_ids;  /* filled while scanning; appended to ids afterwards */
    uint32_t n_plain = 0;               /* ids[0, n_plain): localised by the lean kernel instantiation */
    DevBuf<uint32_t> d_ids;
    uint32_t n_cap = 0, list_cap = 1, pos_cap = 1;
    /* Signatures localised together (winner included) -- LDS per wave decides the occupancy of
     * localize, the number of batches its instruction count. */
    /* the A/B overrides of the handle that owns the plan (PYA_SB, PYA_GTP, PYA_HASH_PP; < 0: the built-in rule), copied in by
     * take_knobs() when the plan is created: a second scorer in the process does not change them (r05 advisor) */
    int knob_sb = -1, knob_gtp = -1, knob_hash_pp = -1;
    void take_knobs(const Knobs &k) {
        knob_sb = k.sb;
        knob_gtp = k.gtp;
        knob_hash_pp = k.hash_pp;
    }
    uint32_t sb() const {
        if (knob_sb >= 0) return (uint32_t)knob_sb;                                /* A/B experiments (PYA_SB) */
        /* one batch holds the winner and one competitor per modified site.  (r01 gave long peptides -- large per-signature
         * tables -- two fewer to keep the LDS, hence the occupancy, up; r05 measured the opposite on cfg3, whose 40-mers then
         * needed two or three batches where one does: 0.465 -> 0.431 ms with the full batch.  These kernels are bound by the
         * instructions they issue, not by residency: DESIGN.md section 6.) */
        uint32_t v = k_max + 1;
        if (v < 2) v = 2;
        if (v > PYA_LOC_SB_MAX) v = PYA_LOC_SB_MAX;
        return v;
    }
    /* Ion types localised per pass (log2): as many as keep one signature's lists <= 256 floats,
     * so long multi-charge lists (cfg4: 228 per type) go one type at a time and the pool -- hence
     * the LDS per wave, hence the occupancy of localize -- stays small. */
    uint32_t gtp() const {
        if (knob_gtp >= 0) return (uint32_t)knob_gtp;                              /* A/B experiments (PYA_GTP) */
        uint32_t g = 0;
        while ((1u << g) < n_types) g++;
        while (g > 0 && (list_cap << g) > 256u) g--;
        return g;
    }
    /* fragment-list slots [signature][type slot][list_cap]: room for sb() signatures */
    uint32_t pool_cap() const {
        const uint32_t per_sig = list_cap << gtp();
        uint32_t want = sb() * per_sig;
        if (want > 2048u) want = 2048u;
        return 2u * per_sig > want ? 2u * per_sig : want;
    }
    uint32_t n_types = 1, k_max = 1;
    uint32_t ns_max = 1;                /* most modifiable residues of one PSM */
    uint32_t z_max = 1;                 /* largest fragment charge in the bucket */
    uint32_t list_max = 1;              /* longest fragment list of one (signature, ion type) */
    uint32_t pair_cap = 1;              /* largest (L - 1) * loss variants: (prefix, variant) pairs of one fragment list */
    uint32_t node_words = 0;            /* largest shared-node shape table (64-bit words) among the PSMs of <= 64 signatures */
    uint32_t node_cols = 0;             /* ... and the most histogram columns one of them needs */
    /* The hash route of the general localize launch (localize_hash.hip.h): ion table for the winner's list and at
     * least one competitor's in-span ions, a grid at most half full, room for the pair lists of a typical PSM
     * (a PSM that needs more is declined and goes to the list-based kernel). */
    uint32_t hash_vc() const { return (2u * list_max + 15u) & ~15u; }
    uint32_t hash_hs() const {
        uint32_t v = 64u;                              /* (a third full at most: 10.15 against 10.45 ms on cfg4 with half) */
        while (v < 3u * hash_vc()) v <<= 1;
        return v;
    }
    /* one direction's pair lists at a time: the winner's and, per competitor of a batch, its in-span pairs on both sides
     * -- room for the worst case, so the hash route never declines for lack of it (PYA_HASH_PP: another multiple of
     * pair_cap, for the tests of the hand-over; 6 instead of 7 measured 4 % slower on cfg4 at the same occupancy: where
     * the arrays behind the lists land in the LDS banks) */
    uint32_t hash_pp() const { return ((knob_hash_pp > 0 ? (uint32_t)knob_hash_pp : 1u + 2u * (sb() - 1u)) * pair_cap + 7u) & ~7u; }
    bool hash_ok(uint32_t max_k, uint32_t n_nl) const {
        return pos_cap <= 64u && hash_vc() <= 8192u &&
               pya_localize_hash_lds_bytes(push_cap(), n_cap, pos_cap, sb(), hash_vc(), hash_hs(), hash_pp(), max_k, n_nl) <= 64u * 1024u;
    }
    uint32_t push_max = 1;              /* largest k * (n_sites - k): single-move competitors of one PSM */
    uint32_t push_cap() const {
        uint32_t v = (push_max + 3u) & ~3u;
        return v > PYA_MAX_PUSHED ? PYA_MAX_PUSHED : v;
    }
};

struct pya_plan {
    pya_handle *h = nullptr;
    uint32_t flags = 0;
    uint64_t n_psm = 0;
    int64_t total_peaks = 0, total_sigs = 0;
    uint32_t peak_cap = 64;
    uint32_t max_k = 1;
    /* host copies needed later */
    std::vector<int64_t> peak_off, sig_off, pep_off, aux_off;
    std::vector<uint32_t> n_sig, order_off;
    std::vector<uint8_t> n_sites, pep;
    std::vector<int32_t> n_of_mod, max_charge;
    /* device metadata */
    DevBuf<int64_t> d_peak_off, d_pep_off, d_aux_off, d_sig_off;
    DevBuf<uint8_t> d_pep, d_n_sites;
    DevBuf<int32_t> d_n_of_mod, d_max_charge, d_status;
    DevBuf<uint32_t> d_aux_pos, d_n_sig, d_order_off, d_ret_n, d_rec, d_sorted;
    DevBuf<float> d_aux_mass, d_ws;
    DevBuf<PeakEntry> d_ret;             /* retained tables, 8-byte entries, every PSM's from an even offset */
    DevBuf<int64_t> d_ret_off;
    std::vector<int64_t> ret_off;        /* [n_psm + 1] */
    DevBuf<uint16_t> d_grid;
    DevBuf<uint32_t> d_redo3;            /* the same for localize's lean instantiation */
    DevBuf<uint32_t> d_redo;             /* [1 + n_psm]: count, then the ids bin_spectra hands to its exact variant */
    Bucket buckets[kNumBuckets];
    /* bin_spectra and score_signatures size their LDS by the peak count, so they are launched per
     * peak class (caps = a few quantiles of the batch's peak counts): one 8 000-peak spectrum must
     * not set the occupancy of a batch of 300-peak spectra.  score lists are additionally split by
     * the C(n,k) class (prefix sharing on / off). */
    struct IdList {
        uint32_t off, n, cap, ncls;
    };
    std::vector<uint32_t> bin_ids, score_ids, fused_ids, big_ids;
    std::vector<IdList> bin_lists, score_lists, big_lists;
    /* launches of the fused kernel: per peak class and charge class, and -- since the kernel's speed
     * follows its LDS footprint -- per LDS class: short peptides with a handful of signatures are not
     * launched with the footprint of the longest peptide with 32 */
    struct FusedLaunch {
        uint32_t off, n, cap, multi_z, n_cap, stride, pos_cap, ent_cap, push_cap;
    };
    std::vector<FusedLaunch> fused_launches;
    uint32_t n_fused_total = 0;
    /* PSMs beyond a limit of the fast kernels (peptide > 64 residues, > 15 000 site assignments, > 2 048 fragments per ion
     * type): binned like every other one, then scored and localised by the general kernel (general_psm.hip) */
    std::vector<uint8_t> gen;           /* [n_psm] */
    std::vector<uint32_t> gen_ids;
    DevBuf<uint32_t> d_gen_ids;
    DevBuf<unsigned char> d_gen_scratch;
    uint32_t gen_l_cap = 1, gen_list_cap = 1;
    std::vector<uint64_t> gen_off;      /* [gen_ids.size() + 1] every general PSM's own slice of the scratch */
    DevBuf<uint64_t> d_gen_off;
    /* ... of them the spectra of more than 8 192 peaks: binned by pya_bin_global_kernel (arrays in the workspace) */
    std::vector<uint32_t> bigbin_ids;
    DevBuf<uint32_t> d_bigbin_ids;
    DevBuf<unsigned char> d_bigbin_scratch;
    uint32_t bigbin_cap = 32;
    size_t bigbin_stride = 0;
    std::vector<uint8_t> big;           /* [n_psm] scored by score_big.hip (thousands of site assignments, plain settings) */
    DevBuf<uint32_t> d_big_ids;
    uint32_t big_pos_cap = 1, big_k_max = 1;
    uint32_t big_kc() const {            /* row length of score_big's count-node table: a power of two >= 8, > the most modifications */
        uint32_t v = 8u;
        while (v < big_k_max + 1u) v <<= 1;
        return v;
    }
    /* score_big's own localisation (summary mode, plain settings, C(n,k) <= pya_big_inline_max()): those PSMs are
     * in no localize list; `bigloc` carries the lean localize body's caps for them, d_redo5 the ones it declines */
    bool big_inline = false;
    uint32_t n_big_inline = 0;
    Bucket bigloc;
    DevBuf<uint32_t> d_redo5;
    /* PSMs scored AND localised by the fused kernel (score_localize.hip): few site assignments, plain
     * settings.  `fusedb` carries the caps the general localize instantiation needs for the ones the
     * fused kernel hands over. */
    Bucket fusedb;
    std::vector<uint8_t> fused;         /* [n_psm] */
    std::vector<uint64_t> desc;         /* [n_psm][PYA_DESC_WORDS] packed descriptors (common.h) */
    DevBuf<uint64_t> d_desc;
    uint32_t fused_both = 0, fused_n_cap = 0, fused_stride = 0, fused_ent_cap = 1;
    DevBuf<uint32_t> d_fused_ids, d_redo4, d_ws_top;
    std::vector<uint8_t> ncls;          /* [n_psm] C(n,k) class of the PSM */
    std::vector<int32_t> pre_status;    /* [n_psm] PSMs the host pre-pass set aside (PYA_FLAG_SKIP_INVALID) */
    uint64_t n_skipped = 0;
    DevBuf<uint32_t> d_bin_ids, d_score_ids;
    /* owned copies of inputs/outputs (pya_score_batch path) */
    DevBuf<double> d_mz, d_inten;
    DevBuf<float> d_best_score, d_ascores;
    DevBuf<uint64_t> d_best_sig, d_alt;
    DevBuf<int32_t> d_n_sig_out;
    DevBuf<unsigned long long> d_stamps;
    DevBuf<unsigned char> arena;          /* one allocation behind every device buffer of the plan */
    /* arena layout: [uploaded metadata (+ spectra) | status (+ results) | workspace]; the middle
     * part comes back to the host in one copy on the pya_score_batch path */
    size_t o_status = 0, d2h_bytes = 0, o_best_score = 0, o_best_sig = 0, o_n_sig_out = 0, o_ascores = 0, o_alt = 0;
    uint32_t io_max_k = 0;
    BatchDev dev;
    /* PYA_FLAG_TIMING: five events per run, in a ring so that a caller can enqueue run after run and read the
     * timings of all of them afterwards (pya_plan_timings_sum) instead of waiting for every run */
    static constexpr uint32_t kEvRing = 128;
    static constexpr uint32_t kEvPerRun = 7;      /* boundaries 0 .. 4 of the caller's stream; 5, 6: begin and end of the fused family on the side stream */
    std::vector<hipEvent_t> evring;      /* [kEvRing][kEvPerRun] */
    std::vector<uint8_t> evalias;        /* [kEvRing][kEvPerRun] the event that marks boundary i of the run: a family that launched
                                          * nothing records no event of its own (a record costs microseconds of stream time);
                                          * entry 5 != 0: the run's fused family ran on the side stream, between events 5 and 6 */
    uint64_t ev_runs = 0, ev_read = 0;   /* runs recorded, runs already summed */
    hipEvent_t *ev_set(uint64_t run) { return evring.data() + kEvPerRun * (run % kEvRing); }
    uint8_t *ev_alias(uint64_t run) { return evalias.data() + kEvPerRun * (run % kEvRing); }
    /* r06 -- a batch of mixed shapes has two independent chains behind the binning: the fused score + localize kernels (few
     * site assignments) and the scoring + localize kernels of everything else.  They run BESIDE each other: the fused family
     * on a stream of the plan's own, forked after the binning and joined at the end of the run (two events); the latency-bound
     * localize kernels and the issue-bound fused kernels fill each other's idle slots, and no launch waits for another
     * family's tail. */
    bool fork = false;
    hipStream_t side = nullptr;          /* the handle's (pya_handle::side_stream: created once, plans share it) */
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t last_stream = nullptr;
    uint64_t n_runs = 0;                 /* pya_plan_run calls so far (which set of hand-over counts is in use) */
    bool ran = false;
    bool quiesced = false;               /* the owner has waited for everything that used the buffers */

    ~pya_plan() {
        for (auto &e : evring)
            if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }
    uint64_t workspace_bytes() const { return arena.bytes(); }
};

/* ---- functions shared between the host files ---- */
/* host_tables.cpp: configuration -> DevConfig, score table, pre-sort order tables */
int build_dev_config(pya_handle *h);
int sync_config(pya_handle *h);
int ensure_lut(pya_handle *h, uint32_t n_max);
uint32_t shape_offset(pya_handle *h, uint32_t n, uint32_t k);
uint32_t next_pow2(uint32_t v);
/* host_plan.cpp: plans (host pre-pass, arena, launches) */
struct IoReq {                       /* pya_score_batch: spectra and results live in the plan's arena too */
    const double *mz, *inten;
    uint32_t max_k;
    double *d_mz_ext, *d_inten_ext;  /* ... unless the caller uploads the spectra itself (big batches) */
    hipStream_t stream;              /* metadata upload: on this stream, waited for alone (nullptr: device-wide) */
    const uint8_t *pre_sites;        /* letter scan already done by the caller: sites per PSM, 255 = invalid letters */
};
void refresh_shared(pya_plan *p);
void fill_dev(pya_plan *p);
int plan_create_impl(pya_handle *h, const pya_batch *b, uint32_t flags, const IoReq *io, pya_plan **out);
int check_status(pya_handle *h, const int32_t *st, uint64_t n, bool skip_invalid = false);
/* host_batch.cpp */
size_t workspace_budget(const pya_handle *h);

#endif
