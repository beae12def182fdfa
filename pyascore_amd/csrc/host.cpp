/* host.cpp -- host side of libpyascore_hip.so: the C ABI of include/pyascore_hip.h.
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
 */
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
size_t pya_score_big_lds_bytes(uint32_t cap, uint32_t pos_cap);
int pya_launch_score_big(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t pos_cap,
                         uint32_t inline_on, hipStream_t stream);
size_t pya_localize_recount_lds_bytes(uint32_t cap, uint32_t push_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb);
size_t pya_localize_hash_lds_bytes(uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t sb, uint32_t vc, uint32_t hs,
                                   uint32_t pp, uint32_t tab_cap, uint32_t max_k, uint32_t n_nl);
int pya_launch_localize_hash(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t push_cap, uint32_t n_cap,
                             uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp, uint32_t vc, uint32_t hs, uint32_t pp,
                             uint32_t tab_cap, uint32_t n_nl, hipStream_t stream);
int pya_launch_localize_recount(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t push_cap,
                                uint32_t pos_cap, uint32_t pool_cap, uint32_t sb, uint32_t gtp, uint32_t *d_redo, hipStream_t stream);
int pya_launch_score_big_list(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max, uint32_t cap,
                              uint32_t pos_cap, hipStream_t stream);
uint32_t pya_big_inline_max(void);
size_t pya_fused_lds_bytes(uint32_t cap, uint32_t n_cap, uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap,
                           uint32_t both, uint32_t multi_z);
int pya_launch_fused(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, uint32_t cap, uint32_t n_cap,
                     uint32_t stride, uint32_t pos_cap, uint32_t ent_cap, uint32_t push_cap, uint32_t both,
                     uint32_t multi_z, uint32_t *d_redo_count, uint32_t *d_redo_ids, hipStream_t stream);
size_t pya_bin_global_scratch_bytes(uint32_t cap);
int pya_launch_bin_global(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch, uint64_t stride,
                          uint32_t cap, hipStream_t stream);
size_t pya_general_lds_bytes(uint32_t l_cap, uint32_t list_cap);
size_t pya_general_scratch_bytes(uint32_t n_cap, uint32_t push_cap);
int pya_launch_general(const BatchDev *b, const uint32_t *d_ids, uint32_t n_ids, unsigned char *d_scratch, uint64_t scratch_stride,
                       uint32_t n_cap, uint32_t push_cap, uint32_t l_cap, uint32_t list_cap, hipStream_t stream);
int pya_launch_localize_redo(const BatchDev *b, const uint32_t *d_count, const uint32_t *d_ids, uint32_t n_max,
                             uint32_t push_cap, uint32_t n_cap, uint32_t pos_cap, uint32_t pool_cap, uint32_t sb,
                             uint32_t gtp, hipStream_t stream);
}

namespace {

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

uint64_t binom(uint32_t n, uint32_t k) {
    if (k > n) return 0;
    if (k > n - k) k = n - k;
    unsigned __int128 r = 1;
    for (uint32_t i = 1; i <= k; i++) {
        r = r * (n - k + i) / i;
        if (r > (unsigned __int128)1 << 62) return ~0ull;
    }
    return (uint64_t)r;
}

bool is_forward(char t) { return t == 'b' || t == 'c'; }
bool is_backward(char t) { return t == 'y' || t == 'z' || t == 'Z'; }

float std_residue_mass(char c) {                       /* Types.h:7-30 */
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

}  // namespace

/* The PYA_* environment switches (route selection for the tests, diagnostics, A/B experiments).  They are
 * read ONCE per handle, in pya_create -- a variable set in a user's shell afterwards changes nothing, and no
 * call pays for getenv -- and again only when pya_reload_env asks for it (the tests flip routes on a live
 * handle that way).  Defaults are the production behaviour. */
struct Knobs {
    bool no_plain = false, no_fused = false, no_big = false, no_tiny = false, no_prefix = false, no_chunks = false;
    bool no_upload_thread = false, one_peak_class = false, peak_classes = false, one_lds_class = false;
    bool host_timing = false, stamps = false, sort_room = false, no_big_inline = false;
    bool no_loc_hash = false, no_nodes = false;
    uint32_t debug = 0;
    int64_t plain_min = 512, big_min_n = 1024, tiny_max = 64;
    uint32_t sort_room_max = 1024;
    int sb = -1, gtp = -1;                      /* < 0: the built-in rule */
    int node_cap = -1;                          /* >= 0: room for that many shared nodes per direction (tests: small values force the walkers) */
    double chunk_mb = 0.;                       /* 0: the default chunk size */
    int64_t workspace_mb = 0;                   /* 0: the default budget */
};
static int g_knob_sb = -1, g_knob_gtp = -1, g_knob_hash_pp = -1;     /* (Bucket has no handle: the two A/B overrides are process-wide) */

static void read_knobs(Knobs &k) {
    auto flag = [](const char *n) { return std::getenv(n) != nullptr; };
    auto num = [](const char *n, int64_t dflt) { const char *v = std::getenv(n); return v ? (int64_t)std::atoll(v) : dflt; };
    k = Knobs();
    k.no_plain = flag("PYA_NO_PLAIN");
    k.no_fused = flag("PYA_NO_FUSED");
    k.no_big = flag("PYA_NO_BIG");
    k.no_tiny = flag("PYA_NO_TINY");
    k.no_prefix = flag("PYA_NO_PREFIX");
    k.no_chunks = flag("PYA_NO_CHUNKS");
    k.no_upload_thread = flag("PYA_NO_UPLOAD_THREAD");
    k.one_peak_class = flag("PYA_ONE_PEAK_CLASS");
    k.peak_classes = flag("PYA_PEAK_CLASSES");
    k.one_lds_class = flag("PYA_ONE_LDS_CLASS");
    k.host_timing = flag("PYA_HOST_TIMING");
    k.stamps = flag("PYA_STAMPS");
    k.sort_room = flag("PYA_SORT_ROOM");
    k.no_big_inline = flag("PYA_NO_BIG_INLINE");
    k.no_loc_hash = flag("PYA_NO_LOC_HASH");
    k.no_nodes = flag("PYA_NO_NODES");
    k.node_cap = (int)num("PYA_NODE_CAP", -1);
    if (const char *d = std::getenv("PYA_DEBUG")) k.debug = (uint32_t)std::strtoul(d, nullptr, 0);
    k.plain_min = num("PYA_PLAIN_MIN", 512);
    k.big_min_n = num("PYA_BIG_MIN_N", 1024);
    k.tiny_max = num("PYA_TINY_MAX", 64);
    k.sort_room_max = (uint32_t)num("PYA_SORT_ROOM_MAX", 1024);
    k.sb = (int)num("PYA_SB", -1);
    k.gtp = (int)num("PYA_GTP", -1);
    if (const char *e = std::getenv("PYA_CHUNK_MB")) k.chunk_mb = std::max(1.0, std::atof(e));
    k.workspace_mb = num("PYA_WORKSPACE_MB", 0);
    g_knob_sb = k.sb;
    g_knob_gtp = k.gtp;
    g_knob_hash_pp = (int)num("PYA_HASH_PP", -1);
}

struct pya_handle {
    Knobs kn;
    int device = 0;
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
        DevBuf<unsigned char> ws;
        uint32_t sig_cap = 0;                      /* signatures the workspace has room for */
        uint32_t seq = 0;
        hipStream_t stream = nullptr;
        BatchDev dev;
        OneMeta meta;                              /* of the last call (pya_rescore_last_keep replays it) */
        bool have_last = false, last_keep = false;
        uint32_t last_max_k = 1;
        pya_plan *view = nullptr;                  /* what pya_get_pep_scores / pya_calculate_ambiguity read */
    } one;

    std::string err;
    int64_t err_index = -1;
    std::vector<int32_t> last_status;         /* per-PSM codes of the last pya_score_batch */
    pya_plan *kept = nullptr;                 /* plan of the last PYA_FLAG_KEEP batch */

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
    uint32_t sb() const {
        if (g_knob_sb >= 0) return (uint32_t)g_knob_sb;                            /* A/B experiments (PYA_SB) */
        /* one batch should hold the winner and one competitor per modified site; long peptides
         * have large per-signature tables, so they get fewer (measured: profiles/r01_c) */
        uint32_t v = k_max + 1;
        if (v < 2) v = 2;
        if (v > PYA_LOC_SB_MAX) v = PYA_LOC_SB_MAX;
        if (pos_cap > 32 && v > 3) v = v - 2 < 3 ? 3 : v - 2;
        return v;
    }
    /* Ion types localised per pass (log2): as many as keep one signature's lists <= 256 floats,
     * so long multi-charge lists (cfg4: 228 per type) go one type at a time and the pool -- hence
     * the LDS per wave, hence the occupancy of localize -- stays small. */
    uint32_t gtp() const {
        if (g_knob_gtp >= 0) return (uint32_t)g_knob_gtp;                          /* A/B experiments (PYA_GTP) */
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
    uint32_t hash_pp() const { return ((g_knob_hash_pp > 0 ? (uint32_t)g_knob_hash_pp : 1u + 2u * (sb() - 1u)) * pair_cap + 7u) & ~7u; }
    bool hash_ok(uint32_t tab_cap, uint32_t max_k, uint32_t n_nl) const {
        return pos_cap <= 64u && hash_vc() <= 8192u &&
               pya_localize_hash_lds_bytes(push_cap(), n_cap, pos_cap, sb(), hash_vc(), hash_hs(), hash_pp(), tab_cap, max_k, n_nl) <= 64u * 1024u;
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
    uint32_t gen_n_cap = 1, gen_push_cap = 1, gen_l_cap = 1, gen_list_cap = 1;
    size_t gen_stride = 0;
    /* ... of them the spectra of more than 8 192 peaks: binned by pya_bin_global_kernel (arrays in the workspace) */
    std::vector<uint32_t> bigbin_ids;
    DevBuf<uint32_t> d_bigbin_ids;
    DevBuf<unsigned char> d_bigbin_scratch;
    uint32_t bigbin_cap = 32;
    size_t bigbin_stride = 0;
    std::vector<uint8_t> big;           /* [n_psm] scored by score_big.hip (thousands of site assignments, plain settings) */
    DevBuf<uint32_t> d_big_ids;
    uint32_t big_pos_cap = 1;
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
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t last_stream = nullptr;
    bool ran = false;
    bool quiesced = false;               /* the owner has waited for everything that used the buffers */

    ~pya_plan() {
        for (auto &e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    uint64_t workspace_bytes() const { return arena.bytes(); }
};

namespace {

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
    if (uniq.size() > PYA_MAX_UNIQ) return h->fail(PYA_ERR_LIMIT, -1, "too many neutral-loss sums");
    c.n_uniq = (int32_t)uniq.size();
    for (size_t j = 0; j < uniq.size(); j++) c.uniq[j] = uniq[j];
    for (int st = 0; st < 256; st++) {
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
void refresh_shared(pya_plan *p) {
    pya_handle *h = p->h;
    BatchDev &d = p->dev;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
}

void fill_dev(pya_plan *p) {
    pya_handle *h = p->h;
    BatchDev &d = p->dev;
    std::memset(&d, 0, sizeof d);
    d.peak_off = p->d_peak_off.p;
    d.pep = p->d_pep.p;
    d.pep_off = p->d_pep_off.p;
    d.n_of_mod = p->d_n_of_mod.p;
    d.max_charge = p->d_max_charge.p;
    d.aux_pos = p->d_aux_pos.p;
    d.aux_mass = p->d_aux_mass.p;
    d.aux_off = p->d_aux_off.p;
    d.n_sites = p->d_n_sites.p;
    d.n_sig = p->d_n_sig.p;
    d.order_off = p->d_order_off.p;
    d.sig_off = p->d_sig_off.p;
    d.desc = p->d_desc.p;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
    d.ret = p->d_ret.p;
    d.ret_off = p->d_ret_off.p;
    d.ret_n = p->d_ret_n.p;
    d.grid = p->d_grid.p;
    d.redo_count = p->d_redo.p;
    d.redo_ids = p->d_redo.p + 64;
    d.redo3_count = p->d_redo3.p;
    d.redo3_ids = p->d_redo3.p + 64;
    d.redo3b_count = p->d_redo3.p + 1;
    d.redo3b_ids = p->d_redo3.p + 64 + p->n_psm;
    d.redo4_count = p->d_redo4.p;
    d.redo4_ids = p->d_redo4.p + 64;
    d.ws = p->d_ws.p;
    d.ws_top = p->d_ws_top.p;
    d.rec = p->d_rec.p;
    d.sorted_idx = p->d_sorted.p;
    d.status = p->d_status.p;
    d.max_k = p->max_k;
    d.keep = (p->flags & PYA_FLAG_KEEP) ? 1u : 0u;
    d.debug = h->kn.debug;
    if (h->kn.stamps) {
        if (!p->d_stamps.p) {
            (void)p->d_stamps.alloc(64);
            (void)hipMemset(p->d_stamps.p, 0, 64 * 8);
        }
        d.stamps = p->d_stamps.p;
    }
}

}  // namespace

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
    delete h;
}

int pya_reload_env(pya_handle *h) {
    if (!h) return PYA_ERR_ARG;
    read_knobs(h->kn);
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

int pya_count_sites(const pya_handle *h, const uint8_t *pep, uint64_t L, int32_t *n_sites, uint8_t *site_pos) {
    if (!h || !pep || !n_sites) return PYA_ERR_ARG;
    int n = 0;
    for (uint64_t i = 0; i < L; i++)
        if (h->letter_modifiable((char)pep[i], i, L)) {
            if (site_pos && n < PYA_MAX_PEPTIDE_LEN) site_pos[n] = (uint8_t)i;
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

namespace {
struct IoReq {                       /* pya_score_batch: spectra and results live in the plan's arena too */
    const double *mz, *inten;
    uint32_t max_k;
    double *d_mz_ext, *d_inten_ext;  /* ... unless the caller uploads the spectra itself (big batches) */
    hipStream_t stream;              /* metadata upload: on this stream, waited for alone (nullptr: device-wide) */
    const uint8_t *pre_sites;        /* letter scan already done by the caller: sites per PSM, 255 = invalid letters */
};
}

static int plan_create_impl(pya_handle *h, const pya_batch *b, uint32_t flags, const IoReq *io, pya_plan **out) {
    if (!h || !b || !out) return PYA_ERR_ARG;
    *out = nullptr;
    h->err.clear();
    h->err_index = -1;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_config(h);
    if (rc) return rc;
    const uint64_t n = b->n_psm;
    if (n > 0 && (!b->peak_off || !b->pep || !b->pep_off || !b->n_of_mod || !b->max_charge))
        return h->fail(PYA_ERR_ARG, -1, "NULL array in batch");
    if (n >= (1ull << 31)) return h->fail(PYA_ERR_LIMIT, -1, "more than 2^31 PSMs in one batch");
    const bool host_timing = h->kn.host_timing;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pya plan] %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    std::unique_ptr<pya_plan> p(new pya_plan);
    p->h = h;
    p->flags = flags;
    p->n_psm = n;
    p->peak_off.assign(b->peak_off, b->peak_off + n + 1);
    p->pep_off.assign(b->pep_off, b->pep_off + n + 1);
    p->n_of_mod.assign(b->n_of_mod, b->n_of_mod + n);
    p->max_charge.assign(b->max_charge, b->max_charge + n);
    const bool has_aux = b->aux_off && b->aux_pos && b->aux_mass;
    if (has_aux) p->aux_off.assign(b->aux_off, b->aux_off + n + 1);
    else p->aux_off.assign(n + 1, 0);
    const int64_t peak_base = n ? p->peak_off[0] : 0, pep_base = n ? p->pep_off[0] : 0,
                  aux_base = n ? p->aux_off[0] : 0;
    p->total_peaks = n ? p->peak_off[n] - peak_base : 0;
    const int64_t total_pep = n ? p->pep_off[n] - pep_base : 0;
    const int64_t total_aux = n ? p->aux_off[n] - aux_base : 0;
    if (p->total_peaks < 0 || total_pep < 0 || total_aux < 0)
        return h->fail(PYA_ERR_ARG, -1, "offset arrays are not monotone (the last offset is below the first)");
    p->pep.assign(b->pep + pep_base, b->pep + pep_base + total_pep);
    for (uint64_t i = 0; i <= n; i++) {
        p->peak_off[i] -= peak_base;
        p->pep_off[i] -= pep_base;
        p->aux_off[i] -= aux_base;
    }
    p->ret_off.resize(n + 1);
    {
        int64_t at = 0;                                     /* every PSM's retained table starts at an even entry */
        for (uint64_t i = 0; i < n; i++) {
            p->ret_off[i] = at;
            const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
            at += ((P > 0 ? P : 0) + 1) & ~(int64_t)1;
        }
        p->ret_off[n] = at;
    }
    lap("copy meta");
    p->n_sites.resize(n);
    p->n_sig.resize(n);
    p->order_off.resize(n);
    p->sig_off.resize(n + 1);
    p->ncls.assign(n, 0);
    const uint32_t n_uniq = (uint32_t)h->cfg.n_uniq, n_types = (uint32_t)h->cfg.n_types;
    uint32_t max_P = 1, lut_need = 0, max_k = 1;
    int64_t sig_total = 0;
    /* (tiny batches are launch-bound: the lean instantiation's extra memset + hand-over launch cost
     * more than its occupancy gains there) */
    const bool plain_on = h->cfg.n_nl == 0 && !(flags & PYA_FLAG_KEEP) && !h->kn.no_plain && n >= (uint64_t)h->kn.plain_min;
    /* Pass A (threaded for big batches): the per-letter work -- validate every PSM and count its
     * modifiable residues.  It only finds the first offending PSM; the detailed message comes from
     * the serial checks below, run for that PSM alone. */
    /* fused score + localize kernel: plain settings with one ion type per direction; C(n,k) <= 32 when
     * both directions are scored (one (signature, direction) walker per lane), <= 64 with one */
    const bool both_dirs = h->cfg.n_fwd > 0 && h->cfg.n_fwd < h->cfg.n_types;
    const bool fused_on = plain_on && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 && !h->kn.no_fused;
    const uint32_t fused_max_n = both_dirs ? 32u : 64u;
    p->fused.assign(n, 0);
    /* score_big.hip: one PSM per 8-wave workgroup, fragment tree shared two levels deep */
    const uint64_t big_min_n = (uint64_t)h->kn.big_min_n;
    const bool big_on = h->cfg.n_nl == 0 && both_dirs && h->cfg.n_fwd == 1 && h->cfg.n_types == 2 && h->mz_error <= 0.49f && !h->kn.no_big;
    p->big.assign(n, 0);
    p->gen.assign(n, 0);
    /* (summary mode with the lean localize route on: the kernel localises what it scores) */
    const bool big_inline_ok = big_on && plain_on && !h->kn.no_big_inline;
    const uint64_t big_inline_max = pya_big_inline_max();
    const bool skip_invalid = (flags & PYA_FLAG_SKIP_INVALID) != 0;
    std::vector<uint8_t> bad(n, 0);
    {
        auto scan = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; i++) {
                const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
                const int64_t L = p->pep_off[i + 1] - p->pep_off[i];
                const int32_t k = p->n_of_mod[i], z = p->max_charge[i];
                bool ok = P > 0 && P <= PYA_MAX_PEAKS && L >= 1 && L <= PYA_MAX_PEPTIDE_LEN && k >= 0 && z >= 1 && z <= 16 &&
                          p->aux_off[i + 1] >= p->aux_off[i];
                uint32_t ns = 0;
                if (ok) {
                    if (io && io->pre_sites) {
                        ns = io->pre_sites[i];
                        ok = ns != 255u;
                    } else {
                        ok = h->scan_peptide(p->pep.data() + p->pep_off[i], L, &ns);
                    }
                    for (int64_t a = p->aux_off[i]; has_aux && a < p->aux_off[i + 1]; a++)
                        ok = ok && b->aux_pos[aux_base + a] <= (uint32_t)L;
                    ok = ok && ns <= PYA_MAX_SITES;
                }
                bad[i] = ok ? 0 : 1;
                p->n_sites[i] = ok ? (uint8_t)ns : 0;
            }
        };
        unsigned nt = n >= 20000 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
        if (nt == 1) {
            scan(0, n);
        } else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; t++) th.emplace_back(scan, n * t / nt, n * (t + 1) / nt);
            for (auto &x : th) x.join();
        }
    }
    lap("letter scan");
    /* an invalid PSM ends the call with its message -- or, with PYA_FLAG_SKIP_INVALID, is set aside
     * (status PYA_ST_INVALID / PYA_ST_OVER_LIMIT, best_score -1, n_sig -1) while the rest is scored */
    p->pre_status.assign(n, 0);
    uint64_t n_skipped = 0;
    auto reject = [&](int code, uint64_t i, const char *fmt, ...) -> int {
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        if (!skip_invalid) return h->fail(code, (int64_t)i, "%s", buf);
        if (n_skipped == 0) (void)h->fail(code, (int64_t)i, "%s", buf);       /* first message is kept */
        n_skipped++;
        p->pre_status[i] = code == PYA_ERR_LIMIT ? PYA_ST_OVER_LIMIT : PYA_ST_INVALID;
        p->n_sites[i] = 0;
        p->n_sig[i] = 0;
        p->order_off[i] = 0;
        p->sig_off[i] = sig_total;
        p->buckets[0].general_ids.push_back((uint32_t)i);     /* localize writes the "no result" record */
        return PYA_OK;
    };
    for (uint64_t i = 0; i < n; i++) {
        const int64_t P = p->peak_off[i + 1] - p->peak_off[i];
        const int64_t L = p->pep_off[i + 1] - p->pep_off[i];
        const int32_t k = p->n_of_mod[i], z = p->max_charge[i];
        uint32_t ns = p->n_sites[i];
        if (bad[i]) {
            int rc2 = PYA_OK;
            const uint64_t iu = i;
            if (P <= 0) rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: empty spectrum", (unsigned long long)iu);
            else if (P > PYA_MAX_PEAKS)
                rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %lld peaks exceed the limit of %d", (unsigned long long)iu,
                             (long long)P, PYA_MAX_PEAKS);
            else if (L < 1 || L > PYA_MAX_PEPTIDE_LEN)
                rc2 = reject(L < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, i, "PSM %llu: peptide length %lld outside 1..%d",
                             (unsigned long long)iu, (long long)L, PYA_MAX_PEPTIDE_LEN);
            else if (k < 0) rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: negative n_of_mod", (unsigned long long)iu);
            else if (z < 1 || z > 16)
                rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: max_fragment_charge %d outside 1..16", (unsigned long long)iu, z);
            else if (p->aux_off[i + 1] < p->aux_off[i])
                rc2 = reject(PYA_ERR_ARG, i, "PSM %llu: aux_off is not monotone", (unsigned long long)iu);
            else {
                const uint8_t *s = p->pep.data() + p->pep_off[i];
                uint32_t cnt = 0;
                int64_t bad_j = -1;
                for (int64_t j = 0; j < L; j++) {
                    if (!h->is_residue[s[j]] && bad_j < 0) bad_j = j;
                    if (h->letter_modifiable((char)s[j], (size_t)j, (size_t)L)) cnt++;
                }
                int64_t bad_a = -1;
                for (int64_t a = p->aux_off[i]; has_aux && a < p->aux_off[i + 1]; a++)
                    if (b->aux_pos[aux_base + a] > (uint32_t)L && bad_a < 0) bad_a = a;
                if (bad_j >= 0)
                    rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: unknown residue '%c' at position %lld", (unsigned long long)iu,
                                 (char)s[bad_j], (long long)(bad_j + 1));
                else if (bad_a >= 0)
                    rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: aux_mod_pos %u beyond the peptide", (unsigned long long)iu,
                                 b->aux_pos[aux_base + bad_a]);
                else if (cnt > PYA_MAX_SITES)
                    rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %u modifiable residues exceed %d", (unsigned long long)iu, cnt,
                                 PYA_MAX_SITES);
                else rc2 = reject(PYA_ERR_PSM, i, "PSM %llu: rejected by the batch scan", (unsigned long long)iu);
            }
            if (rc2) return rc2;
            continue;
        }
        uint64_t N = 0;
        if ((uint32_t)k <= ns) {
            uint64_t &cached = h->binom_cache[ns][k];
            if (cached == 0) cached = binom(ns, (uint32_t)k);
            N = cached;
        }
        if (N > PYA_MAX_SIGNATURES) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: C(%u,%d) site assignments exceed the limit of %d",
                             (unsigned long long)i, ns, k, PYA_MAX_SIGNATURES);
            if (rc2) return rc2;
            continue;
        }
        const uint32_t per_type = (uint32_t)(L - 1) * (uint32_t)z * n_uniq;
        if (per_type > PYA_MAX_FRAGMENTS_PER_TYPE) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: %u fragments per ion type exceed %d", (unsigned long long)i,
                             per_type, PYA_MAX_FRAGMENTS_PER_TYPE);
            if (rc2) return rc2;
            continue;
        }
        if ((uint64_t)per_type * n_types > PYA_MAX_LUT_N) {
            int rc2 = reject(PYA_ERR_LIMIT, i, "PSM %llu: up to %llu theoretical fragments per site assignment; the score table covers %d",
                             (unsigned long long)i, (unsigned long long)per_type * n_types, PYA_MAX_LUT_N);
            if (rc2) return rc2;
            continue;
        }
        p->n_sites[i] = (uint8_t)ns;
        p->n_sig[i] = (uint32_t)N;
        uint32_t ooff = 0;
        if (N) {
            uint32_t &co = h->shape_cache[ns][k];
            if (co == 0xffffffffu) co = shape_offset(h, ns, (uint32_t)k);
            ooff = co;
        }
        p->order_off[i] = ooff;
        p->sig_off[i] = sig_total;
        sig_total += (int64_t)N;
        if (P <= PYA_FAST_PEAKS) max_P = std::max<uint32_t>(max_P, (uint32_t)P);    /* (sizes the LDS of the fast kernels) */
        lut_need = std::max(lut_need, per_type * n_types);
        if ((uint32_t)k > max_k) max_k = (uint32_t)k;
        if (P > PYA_FAST_PEAKS || L > PYA_FAST_PEPTIDE_LEN || N > PYA_FAST_SIGNATURES || per_type > PYA_FAST_FRAGMENTS_PER_TYPE || h->n_top != PYA_NTOP) {
            /* beyond a limit of the fast kernels (or n_top > 10): the general kernel takes the PSM whole */
            p->gen[i] = 1;
            p->gen_ids.push_back((uint32_t)i);
            p->gen_n_cap = std::max<uint32_t>(p->gen_n_cap, (uint32_t)N);
            p->gen_push_cap = std::max<uint32_t>(p->gen_push_cap, (uint32_t)k <= ns ? (uint32_t)k * (ns - (uint32_t)k) : 0u);
            p->gen_l_cap = std::max<uint32_t>(p->gen_l_cap, (uint32_t)L);
            p->gen_list_cap = std::max<uint32_t>(p->gen_list_cap, per_type);
            if (P > PYA_FAST_PEAKS) {
                p->bigbin_ids.push_back((uint32_t)i);
                p->bigbin_cap = std::max<uint32_t>(p->bigbin_cap, ((uint32_t)P + 31u) & ~31u);
            }
            continue;
        }
        int cls_of_i = 0;
        if (N > 0 && (uint32_t)k < ns) {
            int bi = 0;
            while (N > kBucketLimits[bi]) bi++;
            cls_of_i = bi;
            /* (its count records hold the cumulative counts as bytes: at most 255 fragments) */
            const uint32_t frags = (both_dirs ? 2u : 1u) * (uint32_t)(L - 1) * (uint32_t)z;
            const bool to_fused = fused_on && N <= fused_max_n && frags <= 255u;
            bool inl = false;
            if (big_on && z == 1 && N > big_min_n && ns >= 11) { /* (its second level shares ten sites: at least eleven) */
                p->big[i] = 1;
                p->big_pos_cap = std::max(p->big_pos_cap, (uint32_t)(L - 1));
                inl = big_inline_ok && N <= big_inline_max;
            }
            Bucket &bk = to_fused ? p->fusedb : (inl ? p->bigloc : p->buckets[bi]);
            if (inl) p->n_big_inline++;
            if (to_fused) {
                p->fused[i] = 1;
                p->fused_ent_cap = std::max(p->fused_ent_cap, (uint32_t)(L - 1) * (uint32_t)z);
            }
            /* lean localize instantiation: no neutral losses, charge 1, summary mode (it checks the
             * residue masses itself and hands back what it cannot do) */
            if (inl) {
                bk.ids.push_back((uint32_t)i);           /* (p->bigloc: localised by the recounting lean launch) */
            } else if (to_fused || (plain_on && z == 1)) bk.ids.push_back((uint32_t)i);
            else bk.general_ids.push_back((uint32_t)i);
            bk.n_cap = std::max<uint32_t>(bk.n_cap, (uint32_t)N);
            bk.list_cap = std::max<uint32_t>(bk.list_cap, next_pow2(std::max<uint32_t>(per_type, 1)));
            bk.pos_cap = std::max<uint32_t>(bk.pos_cap, (uint32_t)std::max<int64_t>(L - 1, 1));
            bk.n_types = n_types;
            bk.k_max = std::max<uint32_t>(bk.k_max, (uint32_t)k);
            bk.push_max = std::max<uint32_t>(bk.push_max, (uint32_t)k * (ns - (uint32_t)k));
            bk.z_max = std::max<uint32_t>(bk.z_max, (uint32_t)z);
            bk.pair_cap = std::max<uint32_t>(bk.pair_cap, (uint32_t)(L - 1) * n_uniq);
            bk.list_max = std::max<uint32_t>(bk.list_max, per_type);
            if (N <= 64) {
                bk.node_words = std::max<uint32_t>(bk.node_words, 2u * (ns + 1u) * (1u + (uint32_t)((N + 7) / 8)));
                auto sc = h->shape_cols.find(ns << 8 | (uint32_t)k);
                if (sc != h->shape_cols.end()) bk.node_cols = std::max(bk.node_cols, sc->second);
            }
        } else {
            Bucket &bk = p->buckets[0];                 /* unambiguous / empty: cheapest launch */
            bk.general_ids.push_back((uint32_t)i);
            bk.n_cap = std::max<uint32_t>(bk.n_cap, (uint32_t)N);
            /* (its peptide still goes through the score kernel of this class, whose residue table is sized by pos_cap) */
            bk.pos_cap = std::max<uint32_t>(bk.pos_cap, (uint32_t)std::max<int64_t>(L - 1, 1));
        }
        p->ncls[i] = (uint8_t)cls_of_i;
    }
    if (p->n_big_inline) {
        const Bucket &bl = p->bigloc;
        p->big_inline = pya_localize_recount_lds_bytes((max_P + 31u) & ~31u, bl.push_cap(), bl.pos_cap, bl.pool_cap(), bl.sb()) <= 64 * 1024;
        if (!p->big_inline) {
            /* (caps that do not fit what score_big's dead tables leave: the separate localize kernels take them) */
            for (uint64_t i = 0; i < n; i++) {
                if (!p->big[i] || p->pre_status[i] || p->n_sig[i] > big_inline_max) continue;
                Bucket &bk = p->buckets[p->ncls[i]];
                bk.ids.push_back((uint32_t)i);                  /* (these are charge-1 PSMs: the lean list) */
                bk.n_cap = std::max(bk.n_cap, p->n_sig[i]);
                bk.list_cap = std::max(bk.list_cap, bl.list_cap);
                bk.pos_cap = std::max(bk.pos_cap, bl.pos_cap);
                bk.n_types = bl.n_types;
                bk.k_max = std::max(bk.k_max, bl.k_max);
                bk.push_max = std::max(bk.push_max, bl.push_max);
                bk.z_max = std::max(bk.z_max, bl.z_max);
            }
            p->n_big_inline = 0;
            p->bigloc.ids.clear();
        }
    }
    for (Bucket &bk : p->buckets) {
        bk.n_plain = (uint32_t)bk.ids.size();
        bk.ids.insert(bk.ids.end(), bk.general_ids.begin(), bk.general_ids.end());
        bk.general_ids.clear();
        bk.general_ids.shrink_to_fit();
    }
    {
        Bucket &fb = p->fusedb;
        fb.n_plain = (uint32_t)fb.ids.size();
        bool keep_fused = !fb.ids.empty();
        if (keep_fused) {
            p->fused_both = both_dirs ? 1u : 0u;
            p->fused_n_cap = (fb.n_cap + 3u) & ~3u;
            p->fused_stride = (both_dirs ? 2u : 1u) * p->fused_n_cap + 4u;
            const uint32_t cap_all = (max_P + 31u) & ~31u;
            keep_fused = pya_fused_lds_bytes(cap_all, p->fused_n_cap, p->fused_stride, fb.pos_cap, p->fused_ent_cap, fb.push_cap(), p->fused_both, 1u) <= 64 * 1024 &&
                         pya_localize_lds_bytes(fb.push_cap(), fb.n_cap, fb.pos_cap, fb.pool_cap(), fb.sb()) <= kMaxLds;
        }
        if (!keep_fused && !fb.ids.empty()) {               /* (huge spectra) back to the two-kernel route */
            Bucket &b0 = p->buckets[0];
            std::vector<uint32_t> lean, general;                /* charge 1 -> lean localize instantiation */
            for (uint32_t id : fb.ids) (p->max_charge[id] == 1 ? lean : general).push_back(id);
            b0.ids.insert(b0.ids.begin(), lean.begin(), lean.end());
            b0.n_plain += (uint32_t)lean.size();
            b0.ids.insert(b0.ids.end(), general.begin(), general.end());
            b0.n_cap = std::max(b0.n_cap, fb.n_cap);
            b0.list_cap = std::max(b0.list_cap, fb.list_cap);
            b0.pos_cap = std::max(b0.pos_cap, fb.pos_cap);
            b0.n_types = std::max(b0.n_types, fb.n_types);
            b0.k_max = std::max(b0.k_max, fb.k_max);
            b0.push_max = std::max(b0.push_max, fb.push_max);
            b0.z_max = std::max(b0.z_max, fb.z_max);
            fb.ids.clear();
            fb.n_plain = 0;
            std::fill(p->fused.begin(), p->fused.end(), 0);
        }
    }
    lap("psm loop");
    p->n_skipped = n_skipped;
    p->sig_off[n] = sig_total;
    p->total_sigs = sig_total;
    p->max_k = max_k;
    p->peak_cap = (max_P + 31u) & ~31u;
    {
        /* peak classes: the median, 90th and 99th percentile and the maximum of the peak counts,
         * rounded up to 32 (one class for small batches) */
        std::vector<uint32_t> caps;
        if (n >= 2048 && !h->kn.one_peak_class) {
            std::vector<uint32_t> pk(n);
            for (uint64_t i = 0; i < n; i++)
                pk[i] = p->pre_status[i] ? 1u : (uint32_t)(p->peak_off[i + 1] - p->peak_off[i]);
            for (uint32_t id : p->bigbin_ids) pk[id] = 1u;         /* (binned by their own kernel) */
            for (double q : {0.5, 0.9, 0.99}) {
                const size_t at = (size_t)(q * (double)(n - 1));
                std::nth_element(pk.begin(), pk.begin() + at, pk.end());
                caps.push_back((pk[at] + 31u) & ~31u);
            }
        }
        /* classes only pay when the tail is long: every extra launch has its own ramp-up and tail */
        if (!caps.empty() && p->peak_cap < 2 * caps[0] && !h->kn.peak_classes) caps.clear();
        caps.push_back(p->peak_cap);
        std::sort(caps.begin(), caps.end());
        caps.erase(std::unique(caps.begin(), caps.end()), caps.end());
        const size_t nc = caps.size();
        std::vector<uint32_t> cnt_bin(nc, 0), cnt_score(nc * kNumBuckets, 0), cnt_big(nc, 0);
        std::vector<uint8_t> pcls(n);
        for (uint64_t i = 0; i < n; i++) {
            if (p->pre_status[i]) continue;                  /* set aside: neither binned nor scored */
            const uint32_t P = (uint32_t)(p->peak_off[i + 1] - p->peak_off[i]);
            if (P > PYA_FAST_PEAKS) continue;               /* (pya_bin_global_kernel; scored by the general kernel) */
            size_t c = 0;
            while (caps[c] < P) c++;
            pcls[i] = (uint8_t)c;
            cnt_bin[c]++;
            if (p->fused[i] || p->gen[i]) continue;
            if (p->big[i]) cnt_big[c]++;
            else cnt_score[p->ncls[i] * nc + c]++;
        }
        uint32_t off = 0;
        for (size_t c = 0; c < nc; c++) {
            p->bin_lists.push_back({off, 0u, caps[c], 0u});
            off += cnt_bin[c];
        }
        off = 0;
        for (size_t g = 0; g < nc * kNumBuckets; g++) {
            p->score_lists.push_back({off, 0u, caps[g % nc], (uint32_t)(g / nc)});
            off += cnt_score[g];
        }
        uint32_t n_score = off;
        off = 0;
        for (size_t c = 0; c < nc; c++) {
            p->big_lists.push_back({off, 0u, caps[c], 0u});
            off += cnt_big[c];
        }
        p->big_ids.resize(off);
        {
            uint32_t nb = 0;
            for (size_t c = 0; c < nc; c++) nb += cnt_bin[c];
            p->bin_ids.resize(nb);
        }
        p->score_ids.resize(n_score);
        for (uint64_t i = 0; i < n; i++) {
            if (p->pre_status[i]) continue;
            if (p->peak_off[i + 1] - p->peak_off[i] > PYA_FAST_PEAKS) continue;
            {
                pya_plan::IdList &bl = p->bin_lists[pcls[i]];
                p->bin_ids[bl.off + bl.n++] = (uint32_t)i;
            }
            if (p->fused[i] || p->gen[i]) {
                /* (listed below / in gen_ids) */
            } else if (p->big[i]) {
                pya_plan::IdList &gl = p->big_lists[pcls[i]];
                p->big_ids[gl.off + gl.n++] = (uint32_t)i;
            } else {
                pya_plan::IdList &sl = p->score_lists[p->ncls[i] * nc + pcls[i]];
                p->score_ids[sl.off + sl.n++] = (uint32_t)i;
            }
        }
            {
            const uint32_t ndir = both_dirs ? 2u : 1u;
            p->n_fused_total = 0;
            for (uint64_t i = 0; i < n; i++) p->n_fused_total += (p->fused[i] && !p->pre_status[i]) ? 1u : 0u;
            /* ---- launches of the fused kernel ---- */
            struct Item { uint32_t id, group; size_t need; uint32_t n_cap, pos, ent, push; };
            std::vector<Item> items;
            for (uint64_t i = 0; i < n; i++) {
                if (!p->fused[i] || p->pre_status[i]) continue;
                Item it;
                it.id = (uint32_t)i;
                const uint32_t z = (uint32_t)p->max_charge[i], Lm1 = (uint32_t)(p->pep_off[i + 1] - p->pep_off[i] - 1);
                const uint32_t kk = (uint32_t)p->n_of_mod[i], ns = p->n_sites[i];
                it.group = (uint32_t)pcls[i] * 2 + (z > 1 ? 1u : 0u);
                it.n_cap = (p->n_sig[i] + 3u) & ~3u;
                it.pos = std::max(Lm1, 1u);
                it.ent = std::max(Lm1 * z, 1u);
                it.push = std::min<uint32_t>(PYA_MAX_PUSHED, (kk * (ns - kk) + 7u) & ~7u);
                if (it.push < 8) it.push = 8;
                it.need = pya_fused_lds_bytes(caps[pcls[i]], it.n_cap, ndir * it.n_cap + 4, it.pos, it.ent, it.push, p->fused_both, z > 1 ? 1u : 0u);
                items.push_back(it);
            }
            std::sort(items.begin(), items.end(), [](const Item &a, const Item &b2) {
                return a.group != b2.group ? a.group < b2.group : (a.need != b2.need ? a.need < b2.need : a.id < b2.id);
            });
            p->fused_ids.resize(items.size());
            size_t g0 = 0;
            while (g0 < items.size()) {
                size_t g1 = g0;
                while (g1 < items.size() && items[g1].group == items[g0].group) g1++;
                /* LDS classes inside the group: cut at the median and the 85th percentile of the footprint when
                 * that buys at least a fifth of the largest footprint (every launch has its own ramp-up and tail) */
                std::vector<size_t> cuts{g0};
                if (g1 - g0 >= 8192 && !h->kn.one_lds_class) {
                    const size_t need_max = items[g1 - 1].need;
                    for (double q : {0.5, 0.85}) {
                        const size_t at = g0 + (size_t)(q * (double)(g1 - g0));
                        size_t cut = at;
                        while (cut < g1 && items[cut].need == items[at].need) cut++;    /* equal footprints stay together */
                        if (cut < g1 && cut > cuts.back() && items[at].need * 5 <= need_max * 4) cuts.push_back(cut);
                    }
                }
                cuts.push_back(g1);
                for (size_t c = 0; c + 1 < cuts.size(); c++) {
                    pya_plan::FusedLaunch fl = {(uint32_t)cuts[c], (uint32_t)(cuts[c + 1] - cuts[c]), caps[items[g0].group / 2],
                                                items[g0].group & 1u, 4, 4, 1, 1, 8};
                    for (size_t t = cuts[c]; t < cuts[c + 1]; t++) {
                        fl.n_cap = std::max(fl.n_cap, items[t].n_cap);
                        fl.pos_cap = std::max(fl.pos_cap, items[t].pos);
                        fl.ent_cap = std::max(fl.ent_cap, items[t].ent);
                        fl.push_cap = std::max(fl.push_cap, items[t].push);
                        p->fused_ids[t] = items[t].id;
                    }
                    fl.stride = ndir * fl.n_cap + 4;            /* + a spare column for lanes without a walker */
                    p->fused_launches.push_back(fl);
                }
                g0 = g1;
            }
        }
    }
    /* packed descriptors: what a kernel needs to know about a PSM before it can fetch anything else,
     * in one cache line (fetched ahead by the fused kernel) */
    p->desc.resize((size_t)n * PYA_DESC_WORDS);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t *w = &p->desc[(size_t)i * PYA_DESC_WORDS];
        const uint64_t L = (uint64_t)std::max<int64_t>(0, std::min<int64_t>(p->pep_off[i + 1] - p->pep_off[i], 0xffff));
        const uint64_t na = (uint64_t)std::max<int64_t>(0, std::min<int64_t>(p->aux_off[i + 1] - p->aux_off[i], 0xffff));
        w[0] = (uint64_t)p->ret_off[i];
        w[1] = (uint64_t)p->pep_off[i];
        w[2] = (uint64_t)p->sig_off[i];
        w[3] = (uint64_t)p->aux_off[i];
        w[4] = L | na << 16 | (uint64_t)((uint32_t)p->n_of_mod[i] & 0xffffu) << 32 | (uint64_t)p->n_sites[i] << 48 |
               (uint64_t)((uint32_t)p->max_charge[i] & 0xffu) << 56;
        w[5] = (uint64_t)p->n_sig[i] | (uint64_t)p->order_off[i] << 32;
    }
    lap("id lists");
    rc = ensure_lut(h, lut_need);
    if (rc) return rc;
    if (h->order_uploaded != h->order_tab.size() || !h->d_order.p) {
        HIPCHK(h, h->d_order.upload(h->order_tab.data(), h->order_tab.size()));
        HIPCHK(h, h->d_inv.upload(h->inv_tab.data(), h->inv_tab.size()));
        if (!h->d_binom.p) {
            std::vector<uint32_t> bt(64 * 64);
            for (uint32_t pp = 0; pp < 64; pp++)
                for (uint32_t t = 0; t < 64; t++) bt[pp * 64 + t] = (uint32_t)std::min<uint64_t>(binom(pp, t), 0xffffffffull);
            HIPCHK(h, h->d_binom.upload(bt.data(), bt.size()));
        }
        h->order_uploaded = h->order_tab.size();
    }
    if (io && io->max_k < max_k)
        return h->fail(PYA_ERR_ARG, -1, "results.max_k (%u) is smaller than the largest n_of_mod (%u)", io->max_k, max_k);
    for (Bucket &bk : p->buckets) {
        if (bk.ids.empty()) continue;
        size_t need = pya_localize_lds_bytes(bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb());
        if (need > kMaxLds)
            return h->fail(PYA_ERR_LIMIT, (int64_t)bk.ids[0], "LDS budget exceeded (%zu bytes) for the bucket of PSM %u",
                           need, bk.ids[0]);
    }
    if (!p->gen_ids.empty() && pya_general_lds_bytes(p->gen_l_cap, p->gen_list_cap) > kMaxLds)
        return h->fail(PYA_ERR_LIMIT, (int64_t)p->gen_ids[0], "LDS budget exceeded for the general kernel (PSM %u)", p->gen_ids[0]);
    lap("tables");
    /* One device allocation for everything (hipMalloc is ~100 us a call), laid out so that what
     * goes up and what comes back are each one contiguous range. */
    {
        struct Up { size_t off; const void *src; size_t bytes; };
        std::vector<Up> ups;
        size_t total = 0;
        auto reserve = [&](size_t bytes) {
            size_t o = total;
            total += (bytes + 255) & ~(size_t)255;
            return o;
        };
        auto meta = [&](const void *src, size_t bytes) {
            size_t o = reserve(bytes);
            if (src && bytes) ups.push_back({o, src, bytes});
            return o;
        };
        const size_t o_ret_off = meta(p->ret_off.data(), (n + 1) * 8);
        const size_t o_peak_off = meta(p->peak_off.data(), (n + 1) * 8), o_pep_off = meta(p->pep_off.data(), (n + 1) * 8),
                     o_aux_off = meta(p->aux_off.data(), (n + 1) * 8), o_sig_off = meta(p->sig_off.data(), (n + 1) * 8),
                     o_pep = meta(p->pep.data(), p->pep.size()), o_n_sites = meta(p->n_sites.data(), n),
                     o_n_of_mod = meta(p->n_of_mod.data(), n * 4), o_max_charge = meta(p->max_charge.data(), n * 4),
                     o_n_sig = meta(p->n_sig.data(), n * 4), o_order_off = meta(p->order_off.data(), n * 4),
                     o_aux_pos = meta(has_aux ? b->aux_pos + aux_base : nullptr, (size_t)total_aux * 4),
                     o_aux_mass = meta(has_aux ? b->aux_mass + aux_base : nullptr, (size_t)total_aux * 4),
                     o_bin_ids = meta(p->bin_ids.data(), p->bin_ids.size() * 4),
                     o_score_ids = meta(p->score_ids.data(), p->score_ids.size() * 4),
                     o_fused_ids = meta(p->fused_ids.data(), p->fused_ids.size() * 4),
                     o_desc = meta(p->desc.data(), p->desc.size() * 8),
                     o_big_ids = meta(p->big_ids.data(), p->big_ids.size() * 4);
        const size_t o_bigloc_ids = meta(p->bigloc.ids.data(), p->bigloc.ids.size() * 4);
        const size_t o_gen_ids = meta(p->gen_ids.data(), p->gen_ids.size() * 4);
        const size_t o_bigbin_ids = meta(p->bigbin_ids.data(), p->bigbin_ids.size() * 4);
        size_t o_bucket_ids[kNumBuckets];
        for (int i = 0; i < kNumBuckets; i++)
            o_bucket_ids[i] = meta(p->buckets[i].ids.data(), p->buckets[i].ids.size() * 4);
        const bool own_spectra = io && !io->d_mz_ext;
        const size_t o_mz = own_spectra ? meta(io->mz + peak_base, (size_t)p->total_peaks * 8) : 0,
                     o_inten = own_spectra ? meta(io->inten + peak_base, (size_t)p->total_peaks * 8) : 0;
        const size_t h2d_bytes = total;
        p->o_status = reserve(n * 4);
        if (io) {
            const size_t mk = io->max_k;
            p->io_max_k = io->max_k;
            p->o_best_score = reserve(n * 4);
            p->o_best_sig = reserve(n * 8);
            p->o_n_sig_out = reserve(n * 4);
            p->o_ascores = reserve(n * mk * 4);
            p->o_alt = reserve(n * mk * 8);
        }
        p->d2h_bytes = total - p->o_status;
        const size_t o_ret_n = reserve(n * 4),
                     o_ret = reserve((size_t)p->ret_off[n] * sizeof(PeakEntry) + 64),
                     o_grid = reserve(n * PYA_GRID_CELLS * 2), o_redo = reserve((n + 64) * 4), o_redo3 = reserve((2 * n + 64) * 4), o_redo4 = reserve(((size_t)p->n_fused_total + 64) * 4), o_redo5 = reserve(((size_t)p->n_big_inline + 64) * 4), o_ws_top = reserve(n * 16),
                     o_ws = reserve((size_t)sig_total * 4), o_rec = reserve((size_t)sig_total * h->rec_words() * 4),
                     o_sorted = reserve((flags & PYA_FLAG_KEEP) ? (size_t)sig_total * 4 : 0);
        p->gen_push_cap = (p->gen_push_cap + 3u) & ~3u;
        p->gen_stride = p->gen_ids.empty() ? 0 : (pya_general_scratch_bytes(p->gen_n_cap, p->gen_push_cap) + 255) & ~(size_t)255;
        const size_t o_gen_scratch = reserve(p->gen_ids.size() * p->gen_stride);
        p->bigbin_stride = p->bigbin_ids.empty() ? 0 : (pya_bin_global_scratch_bytes(p->bigbin_cap) + 255) & ~(size_t)255;
        const size_t o_bigbin_scratch = reserve(p->bigbin_ids.size() * p->bigbin_stride);
        if (!p->arena.take_if_fits(h->spare_arena, total) && !p->arena.take_if_fits(h->spare_arena2, total))
            HIPCHK(h, p->arena.alloc(total));
        unsigned char *base = p->arena.p;
        p->d_peak_off.adopt(base + o_peak_off, n + 1);
        p->d_pep_off.adopt(base + o_pep_off, n + 1);
        p->d_aux_off.adopt(base + o_aux_off, n + 1);
        p->d_sig_off.adopt(base + o_sig_off, n + 1);
        p->d_pep.adopt(base + o_pep, p->pep.size());
        p->d_n_sites.adopt(base + o_n_sites, n);
        p->d_n_of_mod.adopt(base + o_n_of_mod, n);
        p->d_max_charge.adopt(base + o_max_charge, n);
        p->d_n_sig.adopt(base + o_n_sig, n);
        p->d_order_off.adopt(base + o_order_off, n);
        p->d_aux_pos.adopt(base + o_aux_pos, (size_t)total_aux);
        p->d_aux_mass.adopt(base + o_aux_mass, (size_t)total_aux);
        p->d_bin_ids.adopt(base + o_bin_ids, p->bin_ids.size());
        p->d_score_ids.adopt(base + o_score_ids, p->score_ids.size());
        p->d_fused_ids.adopt(base + o_fused_ids, p->fused_ids.size());
        p->d_redo5.adopt(base + o_redo5, (size_t)p->n_big_inline + 64);
        p->d_desc.adopt(base + o_desc, p->desc.size());
        p->d_big_ids.adopt(base + o_big_ids, p->big_ids.size());
        for (int i = 0; i < kNumBuckets; i++)
            p->buckets[i].d_ids.adopt(base + o_bucket_ids[i], p->buckets[i].ids.size());
        p->bigloc.d_ids.adopt(base + o_bigloc_ids, p->bigloc.ids.size());
        p->d_gen_ids.adopt(base + o_gen_ids, p->gen_ids.size());
        p->d_gen_scratch.adopt(base + o_gen_scratch, p->gen_ids.size() * p->gen_stride);
        p->d_bigbin_ids.adopt(base + o_bigbin_ids, p->bigbin_ids.size());
        p->d_bigbin_scratch.adopt(base + o_bigbin_scratch, p->bigbin_ids.size() * p->bigbin_stride);
        if (io) {
            if (own_spectra) {
                p->d_mz.adopt(base + o_mz, (size_t)p->total_peaks);
                p->d_inten.adopt(base + o_inten, (size_t)p->total_peaks);
            } else {
                p->d_mz.adopt(io->d_mz_ext, (size_t)p->total_peaks);
                p->d_inten.adopt(io->d_inten_ext, (size_t)p->total_peaks);
            }
            p->d_best_score.adopt(base + p->o_best_score, n);
            p->d_best_sig.adopt(base + p->o_best_sig, n);
            p->d_n_sig_out.adopt(base + p->o_n_sig_out, n);
            p->d_ascores.adopt(base + p->o_ascores, n * io->max_k);
            p->d_alt.adopt(base + p->o_alt, n * io->max_k);
        }
        p->d_status.adopt(base + p->o_status, n);
        p->d_ret_n.adopt(base + o_ret_n, n);
        p->d_ret.adopt(base + o_ret, (size_t)p->ret_off[n] + 8);
        p->d_ret_off.adopt(base + o_ret_off, n + 1);
        p->d_grid.adopt(base + o_grid, n * PYA_GRID_CELLS);
        p->d_redo.adopt(base + o_redo, n + 64);
        p->d_redo3.adopt(base + o_redo3, 2 * n + 64);
        p->d_redo4.adopt(base + o_redo4, (size_t)p->n_fused_total + 64);
        p->d_ws_top.adopt(base + o_ws_top, n * 4);
        p->d_ws.adopt(base + o_ws, (size_t)sig_total);
        p->d_rec.adopt(base + o_rec, (size_t)sig_total * h->rec_words());
        if (flags & PYA_FLAG_KEEP) p->d_sorted.adopt(base + o_sorted, (size_t)sig_total);
        if (h2d_bytes <= kStageLimit) {
            /* small batch: HIP call overhead dominates, so gather on the host and copy once */
            h->stage.resize(std::max(h->stage.size(), h2d_bytes));
            for (const Up &u : ups) std::memcpy(h->stage.data() + u.off, u.src, u.bytes);
            HIPCHK(h, hipMemcpy(base, h->stage.data(), h2d_bytes, hipMemcpyHostToDevice));
        } else {
            hipStream_t ust = io ? io->stream : nullptr;
            for (const Up &u : ups) HIPCHK(h, hipMemcpyAsync(base + u.off, u.src, u.bytes, hipMemcpyHostToDevice, ust));
            if (ust) HIPCHK(h, hipStreamSynchronize(ust));
            else HIPCHK(h, hipDeviceSynchronize());
        }
    }
    if (n_skipped)      /* bin_spectra never touches these entries, so they keep their code for every run */
        HIPCHK(h, hipMemcpy(p->d_status.p, p->pre_status.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    lap("arena+upload");
    if (flags & PYA_FLAG_TIMING)
        for (auto &e : p->ev) HIPCHK(h, hipEventCreate(&e));
    fill_dev(p.get());
    *out = p.release();
    return PYA_OK;
}

int pya_plan_create(pya_handle *h, const pya_batch *b, uint32_t flags, pya_plan **out) {
    return plan_create_impl(h, b, flags, nullptr, out);
}

int pya_plan_run(pya_plan *p, const double *d_mz, const double *d_inten, void *hip_stream,
                 const pya_results *o) {
    if (!p || !o) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (p->n_psm == 0) return PYA_OK;
    if (!d_mz || !d_inten || !o->best_score || !o->best_sig || !o->n_sig || !o->ascores || !o->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL device pointer passed to pya_plan_run");
    if (o->max_k < p->max_k)
        return h->fail(PYA_ERR_ARG, -1, "results.max_k (%u) is smaller than the largest n_of_mod (%u)",
                       o->max_k, p->max_k);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)hip_stream;
    refresh_shared(p);
    BatchDev d = p->dev;
    d.mz = d_mz;
    d.inten = d_inten;
    d.best_score = o->best_score;
    d.best_sig = o->best_sig;
    d.n_sig_out = o->n_sig;
    d.ascores = o->ascores;
    d.alt_mask = o->alt_mask;
    d.max_k = o->max_k;
    const bool timing = p->flags & PYA_FLAG_TIMING;
    /* a handful of PSMs (PyAscore.score is a batch of one) is launch-bound: one fused launch, one
     * wavefront per PSM, instead of the five of the three-kernel path (tiny_batch.hip) */
    const uint64_t tiny_max = (uint64_t)h->kn.tiny_max;
    bool tiny = !timing && p->n_psm <= tiny_max && p->n_skipped == 0 && !h->kn.no_tiny && p->gen_ids.empty();
    Bucket m;                                               /* caps that cover every PSM of the batch */
    uint32_t prefix = 0, compact = 0;
    if (tiny) {
        /* (the PSMs the fused kernel would take are accounted in their own bucket: its caps count too --
         * leaving them out sized this launch's work areas for the other PSMs only) */
        std::vector<const Bucket *> all;
        for (const Bucket &bk : p->buckets) all.push_back(&bk);
        all.push_back(&p->fusedb);
        all.push_back(&p->bigloc);
        for (const Bucket *pbk : all) {
            const Bucket &bk = *pbk;
            if (bk.ids.empty() && pbk != &p->bigloc) continue;
            if (pbk == &p->bigloc && p->n_big_inline == 0) continue;
            m.n_cap = std::max(m.n_cap, bk.n_cap);
            m.list_cap = std::max(m.list_cap, bk.list_cap);
            m.pos_cap = std::max(m.pos_cap, bk.pos_cap);
            m.n_types = std::max(m.n_types, bk.n_types);
            m.k_max = std::max(m.k_max, bk.k_max);
            m.push_max = std::max(m.push_max, bk.push_max);
            m.z_max = std::max(m.z_max, bk.z_max);
        }
        prefix = (m.n_cap >= 128 && !h->kn.no_prefix) ? 1u : 0u;
        compact = (h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 && m.z_max == 1) ? 1u : 0u;
        /* the merged caps (maxima over the buckets) can ask for more LDS than any single bucket does:
         * such a batch takes the three-kernel path, whose launches are sized per bucket */
        if (pya_tiny_lds_bytes(p->peak_cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, m.push_cap(), m.n_cap, m.pos_cap,
                               m.pool_cap(), m.sb()) > kMaxLds)
            tiny = false;
    }
    if (tiny) {
        int e = pya_launch_tiny(&d, (uint32_t)p->n_psm, p->peak_cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact,
                                m.push_cap(), m.n_cap, m.pos_cap, m.pool_cap(), m.sb(), m.gtp(), st);
        if (e) return h->hip_fail((hipError_t)e, "tiny_batch launch");
        p->last_stream = st;
        p->ran = true;
        p->dev = d;
        return PYA_OK;
    }
    if (timing) HIPCHK(h, hipEventRecord(p->ev[0], st));
    HIPCHK(h, hipMemsetAsync(d.redo_count, 0, sizeof(uint32_t), st));
    int e = 0;
    for (const pya_plan::IdList &l : p->bin_lists) {
        e = pya_launch_bin(&d, p->d_bin_ids.p + l.off, l.n, l.cap, st);
        if (e) return h->hip_fail((hipError_t)e, "bin_spectra launch");
    }
    e = pya_launch_bin_exact(&d, (uint32_t)p->n_psm, p->peak_cap, st);
    if (e) return h->hip_fail((hipError_t)e, "bin_spectra (exact) launch");
    e = pya_launch_bin_global(&d, p->d_bigbin_ids.p, (uint32_t)p->bigbin_ids.size(), p->d_bigbin_scratch.p, p->bigbin_stride, p->bigbin_cap, st);
    if (e) return h->hip_fail((hipError_t)e, "bin_spectra (global) launch");
    if (timing) HIPCHK(h, hipEventRecord(p->ev[1], st));
    for (const pya_plan::IdList &l : p->score_lists) {
        /* classes with C(n,k) > 64 share the walk over the first sites between signatures */
        const uint32_t prefix = (p->buckets[l.ncls].n_cap >= 128 && !h->kn.no_prefix) ? 1u : 0u;
        /* every PSM of the class on the straight-line walker: compact prefix entries */
        const uint32_t compact = (h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1 &&
                                  p->buckets[l.ncls].z_max == 1) ? 1u : 0u;
        /* general settings (neutral losses, several ion types per direction): one lookup set per distinct node of
         * the assignment tree instead of one per signature (score_core.hip.h: score_nodes_dir) */
        const Bucket &sbk = p->buckets[l.ncls];
        const bool general = h->cfg.n_nl != 0 || h->cfg.n_fwd > 1 || h->cfg.n_types - h->cfg.n_fwd > 1;
        uint32_t node_cap = 0, node_cols = std::max<uint32_t>(8u, (sbk.node_cols + 7u) & ~7u);
        /* (the node kernel's LDS decides its occupancy: residue and loss-state tables by the launch, room for 320 nodes
         * per direction -- cfg4's shape needs 186 on average, 328 at most; a direction with more is walked) */
        const uint32_t res_cap = std::min<uint32_t>(64u, (sbk.pos_cap + 1u + 3u) & ~3u);
        const uint32_t nnl_s = (uint32_t)h->cfg.n_nl, nl_cap = nnl_s >= 4u ? 256u : (nnl_s == 0u ? 4u : 1u << (2u * nnl_s));
        if (general && !prefix && !h->kn.no_nodes && sbk.node_words) {
            node_cap = std::min<uint32_t>(320u, (sbk.pos_cap * std::min<uint32_t>(sbk.n_cap, 64u) + 1u) & ~1u);
            if (h->kn.node_cap >= 0) node_cap = (uint32_t)h->kn.node_cap & ~1u;
            if (pya_score_node_lds_bytes(l.cap, h->cfg.n_nl != 0 ? 1u : 0u, node_cap, node_cols, sbk.node_words, res_cap, nl_cap) > 64u * 1024u)
                node_cap = 0;
        }
        e = pya_launch_score(&d, p->d_score_ids.p + l.off, l.n, l.cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, node_cap, node_cols,
                             sbk.node_words, res_cap, nl_cap, st);
        if (e) return h->hip_fail((hipError_t)e, "score_signatures launch");
    }
    for (const pya_plan::IdList &l : p->big_lists) {
        e = pya_launch_score_big(&d, p->d_big_ids.p + l.off, l.n, l.cap, p->big_pos_cap, p->big_inline ? 1u : 0u, st);
        if (e) return h->hip_fail((hipError_t)e, "score_big launch");
    }
    if (timing) HIPCHK(h, hipEventRecord(p->ev[2], st));
    if (p->n_fused_total) {
        /* few site assignments, plain settings: scored and localised in one pass (score_localize.hip), one PSM per
         * wavefront; what it hands over goes through the general localize instantiation */
        HIPCHK(h, hipMemsetAsync(d.redo4_count, 0, sizeof(uint32_t), st));
        const Bucket &fb = p->fusedb;
        for (const pya_plan::FusedLaunch &l : p->fused_launches) {
            e = pya_launch_fused(&d, p->d_fused_ids.p + l.off, l.n, l.cap, l.n_cap, l.stride, l.pos_cap, l.ent_cap, l.push_cap,
                                 p->fused_both, l.multi_z, d.redo4_count, d.redo4_ids, st);
            if (e) return h->hip_fail((hipError_t)e, "score_localize launch");
        }
        e = pya_launch_localize_redo(&d, d.redo4_count, d.redo4_ids, p->n_fused_total, fb.push_cap(), fb.n_cap,
                                     fb.pos_cap, fb.pool_cap(), fb.sb(), fb.gtp(), st);
        if (e) return h->hip_fail((hipError_t)e, "localize (hand-over) launch");
    }
    if (timing) HIPCHK(h, hipEventRecord(p->ev[3], st));
    if (p->big_inline && !p->bigloc.ids.empty()) {
        /* what score_big scored in its summary mode: the lean body with recounted signatures and the winner score_big
         * named; what that declines is scored again with count records and goes to the general localize body */
        const Bucket &bl = p->bigloc;
        HIPCHK(h, hipMemsetAsync(p->d_redo5.p, 0, sizeof(uint32_t), st));
        e = pya_launch_localize_recount(&d, bl.d_ids.p, (uint32_t)bl.ids.size(), 0u, bl.push_cap(), bl.pos_cap, bl.pool_cap(),
                                        bl.sb(), bl.gtp(), p->d_redo5.p, st);
        if (e) return h->hip_fail((hipError_t)e, "localize (recount) launch");
        e = pya_launch_score_big_list(&d, p->d_redo5.p, p->d_redo5.p + 64, p->n_big_inline, p->peak_cap, p->big_pos_cap, st);
        if (e) return h->hip_fail((hipError_t)e, "score_big (hand-over) launch");
        e = pya_launch_localize_redo(&d, p->d_redo5.p, p->d_redo5.p + 64, p->n_big_inline, bl.push_cap(), (uint32_t)pya_big_inline_max(),
                                     bl.pos_cap, bl.pool_cap(), bl.sb(), bl.gtp(), st);
        if (e) return h->hip_fail((hipError_t)e, "localize (score_big hand-over) launch");
    }
    for (Bucket &bk : p->buckets) {
        /* more than sort_room_max signatures: the lean launch without room for the sort emulation (LDS ->
         * occupancy); PSMs with a tie at the top go through the hand-over list to a second lean pass that has it */
        const uint32_t sort_room = (bk.n_cap <= h->kn.sort_room_max || h->kn.sort_room) ? 1u : 0u;
        e = pya_launch_localize(&d, bk.d_ids.p, bk.n_plain, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb(),
                                bk.gtp(), 1u, sort_room, st);
        if (e) return h->hip_fail((hipError_t)e, "localize launch");
        const uint32_t nnl = (uint32_t)h->cfg.n_nl;
        const uint32_t tab_cap = 0u;
        if (!h->kn.no_loc_hash && bk.hash_ok(tab_cap, p->max_k, nnl))
            e = pya_launch_localize_hash(&d, bk.d_ids.p + bk.n_plain, (uint32_t)bk.ids.size() - bk.n_plain, bk.push_cap(), bk.n_cap,
                                         bk.pos_cap, bk.pool_cap(), bk.sb(), bk.gtp(), bk.hash_vc(), bk.hash_hs(), bk.hash_pp(), tab_cap,
                                         nnl, st);
        else
        e = pya_launch_localize(&d, bk.d_ids.p + bk.n_plain, (uint32_t)bk.ids.size() - bk.n_plain, bk.push_cap(), bk.n_cap,
                                bk.pos_cap, bk.pool_cap(), bk.sb(), bk.gtp(), 0u, 1u, st);
        if (e) return h->hip_fail((hipError_t)e, "localize launch");
    }
    if (!p->gen_ids.empty()) {
        e = pya_launch_general(&d, p->d_gen_ids.p, (uint32_t)p->gen_ids.size(), p->d_gen_scratch.p, p->gen_stride, p->gen_n_cap,
                               p->gen_push_cap, p->gen_l_cap, p->gen_list_cap, st);
        if (e) return h->hip_fail((hipError_t)e, "general kernel launch");
    }
    if (timing) HIPCHK(h, hipEventRecord(p->ev[4], st));
    p->last_stream = st;
    p->ran = true;
    p->dev = d;
    return PYA_OK;
}

int pya_plan_timings(pya_plan *p, float ms[4]) {
    if (!p || !ms) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (!(p->flags & PYA_FLAG_TIMING) || !p->ran) return h->fail(PYA_ERR_STATE, -1, "plan has no timing events");
    HIPCHK(h, hipEventSynchronize(p->ev[4]));
    for (int i = 0; i < 4; i++) HIPCHK(h, hipEventElapsedTime(&ms[i], p->ev[i], p->ev[i + 1]));
    return PYA_OK;
}

static int check_status(pya_handle *h, const int32_t *st, uint64_t n, bool skip_invalid = false) {
    if (skip_invalid) return PYA_OK;                    /* codes are reported per PSM instead */
    for (uint64_t i = 0; i < n; i++) {
        switch (st[i]) {
            case PYA_ST_OK: break;
            case PYA_ST_INVALID:
            case PYA_ST_OVER_LIMIT:
                return h->fail(st[i] == PYA_ST_INVALID ? PYA_ERR_PSM : PYA_ERR_LIMIT, (int64_t)i,
                               "PSM %llu was set aside by the host pre-pass", (unsigned long long)i);
            case PYA_ST_NO_BINS:
                return h->fail(PYA_ERR_PSM, (int64_t)i, "PSM %llu: all peaks sit on one multiple of 100 m/z; the "
                               "spectrum has no windows", (unsigned long long)i);
            case PYA_ST_TOO_MANY_BINS:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: more than 65535 m/z windows", (unsigned long long)i);
            case PYA_ST_LUT_RANGE:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: trial count outside the score table", (unsigned long long)i);
            case PYA_ST_PUSHED_OVERFLOW:
                return h->fail(PYA_ERR_LIMIT, (int64_t)i, "PSM %llu: more than %d tied competitors", (unsigned long long)i, PYA_MAX_PUSHED);
            default:
                return h->fail(PYA_ERR_HIP, (int64_t)i, "PSM %llu: unexpected kernel status %d", (unsigned long long)i, st[i]);
        }
    }
    return PYA_OK;
}

int pya_plan_check(pya_plan *p) {
    if (!p) return PYA_ERR_ARG;
    pya_handle *h = p->h;
    if (!p->ran || p->n_psm == 0) return PYA_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(p->last_stream));
    std::vector<int32_t> st(p->n_psm);
    HIPCHK(h, hipMemcpy(st.data(), p->d_status.p, p->n_psm * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h->kn.host_timing) {                               /* diagnostics: how many PSMs the lean kernels handed over */
        uint32_t r3 = 0, r4 = 0;
        (void)hipMemcpy(&r3, p->d_redo3.p, 4, hipMemcpyDeviceToHost);
        if (p->d_redo4.p) (void)hipMemcpy(&r4, p->d_redo4.p, 4, hipMemcpyDeviceToHost);
        std::fprintf(stderr, "[pya plan] handed over: %u by the lean localize instantiation (last bucket), %u of %u by the fused kernel\n",
                     r3, r4, p->n_fused_total);
    }
    const bool skip = (p->flags & PYA_FLAG_SKIP_INVALID) != 0;
    if (skip) h->last_status = st;
    return check_status(h, st.data(), p->n_psm, skip);
}

uint64_t pya_plan_workspace_bytes(const pya_plan *p) { return p ? p->workspace_bytes() : 0; }
uint64_t pya_plan_total_signatures(const pya_plan *p) { return p ? (uint64_t)p->total_sigs : 0; }

void pya_plan_destroy(pya_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->h->device);
    if (p->d_stamps.p) {
        unsigned long long v[64];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(v, p->d_stamps.p, sizeof v, hipMemcpyDeviceToHost);
        unsigned long long tot = 0;
        for (int i = 0; i < 64; i++) tot += v[i];
        std::fprintf(stderr, "[pya stamps] total %llu\n", tot);
        for (int i = 0; i < 64; i++)
            if (v[i]) std::fprintf(stderr, "[pya stamps] phase %2d: %12llu  %5.1f%%\n", i, v[i], 100.0 * v[i] / tot);
    }
    if (p->h->kept == p) p->h->kept = nullptr;
    if (p->h->one.view == p) p->h->one.view = nullptr;       /* (pya_score_one's retained view of its workspace) */
    if (!p->quiesced) (void)hipDeviceSynchronize();     /* nothing may still be using the buffers */
    if (!p->h->spare_arena.p) p->arena.give_to(p->h->spare_arena);
    else if (!p->h->spare_arena2.p) p->arena.give_to(p->h->spare_arena2);
    else if (p->h->spare_arena.n <= p->h->spare_arena2.n) p->arena.give_to(p->h->spare_arena);
    else p->arena.give_to(p->h->spare_arena2);
    delete p;
}

namespace {

const size_t kChunkMin = 32u << 20;          /* spectra bytes below which a call is not worth pipelining */
const size_t kChunkTarget = 96u << 20;       /* spectra bytes per chunk when the budget allows more       */
const size_t kDefaultBudget = (size_t)6 << 30;

size_t workspace_budget(const pya_handle *h) {
    if (h->ws_budget) return h->ws_budget;
    if (h->kn.workspace_mb > 0) return (size_t)std::max<int64_t>(16, h->kn.workspace_mb) << 20;
    return kDefaultBudget;
}

/* Device bytes a chunk [lo, hi) holds while it is scored: its spectra in the upload ring (two
 * slots, so twice) and its arena (retained table, grid, per-signature scores and records, results
 * and metadata).  C(n,k) comes from the peptide letters, as in the plan's pre-pass. */
struct ChunkCost {
    std::vector<double> arena, io;             /* per PSM */
    std::vector<uint8_t> sites;                /* modifiable residues per PSM, 255 = invalid letters / length */
};
ChunkCost chunk_costs(pya_handle *h, const pya_batch *b, uint32_t max_k) {
    const uint64_t n = b->n_psm;
    ChunkCost c;
    c.arena.resize(n);
    c.io.resize(n);
    c.sites.assign(n, 255);
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; i++) {
            const int64_t P = std::max<int64_t>(0, b->peak_off[i + 1] - b->peak_off[i]);
            const int64_t L = b->pep_off[i + 1] - b->pep_off[i];
            double sigs = 0;
            if (L >= 1 && L <= PYA_MAX_PEPTIDE_LEN) {
                uint32_t ns = 0;
                const bool ok = h->scan_peptide(b->pep + b->pep_off[i], L, &ns);
                if (ok && ns < 255u) c.sites[i] = (uint8_t)ns;
                uint64_t N = 0;
                if (ok && ns <= PYA_MAX_SITES && b->n_of_mod[i] >= 0 && (uint32_t)b->n_of_mod[i] <= ns) {
                    uint64_t &cached = h->binom_cache[ns][b->n_of_mod[i]];      /* benign race: same value */
                    if (cached == 0) cached = binom(ns, (uint32_t)b->n_of_mod[i]);
                    N = cached;
                }
                sigs = N > PYA_MAX_SIGNATURES ? 0. : (double)N;
            }
            c.io[i] = 16.0 * (double)P;
            c.arena[i] = 8.0 * (double)(P + 1) + 8.0 + 28.0 * sigs + (double)L + 2.0 * PYA_GRID_CELLS + 96.0 + 12.0 * max_k;
        }
    };
    unsigned nt = n >= 20000 ? std::min(8u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto &x : th) x.join();
    }
    return c;
}

/* "PSM 12: ..." of a chunk that starts at PSM `lo` of the caller's batch -> "PSM <12 + lo>: ..." */
void rebase_error(pya_handle *h, uint64_t lo) {
    if (h->err_index >= 0) h->err_index += (int64_t)lo;
    unsigned long long local = 0;
    int used = 0;
    if (lo && std::sscanf(h->err.c_str(), "PSM %llu%n", &local, &used) == 1)
        h->err = "PSM " + std::to_string(local + lo) + h->err.substr((size_t)used);
}

}  // namespace

/* Big pya_score_batch calls: the batch is cut into chunks of consecutive PSMs that fit the device
 * budget and the chunks are pipelined -- a helper thread streams the spectra of chunk c + 1 over
 * PCIe (the bound of this entry point: 16 bytes per peak) into the other slot of a two-slot ring
 * while this thread plans chunk c, runs its kernels and brings its results back on a second
 * stream.  A call of any size completes; it never fails for lack of workspace. */
static int score_batch_chunked(pya_handle *h, const pya_batch *b, const double *mz, const double *inten,
                               uint32_t flags, const pya_results *out, const std::vector<uint64_t> &cuts,
                               const uint8_t *pre_sites) {
    const size_t nchunk = cuts.size() - 1;
    const uint32_t mk = out->max_k;
    const bool skip = (flags & PYA_FLAG_SKIP_INVALID) != 0;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    if (!h->run_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->run_stream, hipStreamNonBlocking));
    size_t slot_peaks = 0;
    for (size_t c = 0; c < nchunk; c++)
        slot_peaks = std::max<size_t>(slot_peaks, (size_t)(b->peak_off[cuts[c + 1]] - b->peak_off[cuts[c]]));
    for (auto &slot : h->io_ring)
        if (slot.n < slot_peaks * 2) HIPCHK(h, slot.alloc(slot_peaks * 2));
    if (skip) h->last_status.assign(b->n_psm, 0);

    /* uploader: chunk c may be written once chunk c - 2 has been consumed */
    std::mutex mu;
    std::condition_variable cv;
    size_t uploaded = 0, consumed = 0;
    bool stop = false;
    hipError_t up_err = hipSuccess;
    const int device = h->device;
    std::thread uploader([&]() {
        hipError_t e = hipSetDevice(device);
        for (size_t c = 0; c < nchunk && e == hipSuccess; c++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || c < consumed + 2; });
                if (stop) break;
            }
            const int64_t p0 = b->peak_off[cuts[c]], np = b->peak_off[cuts[c + 1]] - p0;
            double *dst = h->io_ring[c & 1].p;
            if (np > 0) {
                e = hipMemcpyAsync(dst, mz + p0, (size_t)np * 8, hipMemcpyHostToDevice, h->copy_stream);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(dst + np, inten + p0, (size_t)np * 8, hipMemcpyHostToDevice, h->copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(h->copy_stream);
            }
            std::lock_guard<std::mutex> lk(mu);
            up_err = e;
            uploaded = c + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess) up_err = e;
        uploaded = nchunk;                                   /* nobody waits for chunks that will not come */
        cv.notify_all();
    });
    auto finish = [&](int rc) {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        uploader.join();
        return rc;
    };

    /* this thread: plan chunk c + 1 (host pre-pass) while the GPU runs the kernels and the result copy of
     * chunk c; two plans alive at a time */
    auto make_plan = [&](size_t c, pya_plan **pp) -> int {
        const uint64_t lo = cuts[c], hi = cuts[c + 1];
        const int64_t np = b->peak_off[hi] - b->peak_off[lo];
        pya_batch sub = *b;
        sub.n_psm = hi - lo;
        sub.peak_off = b->peak_off + lo;
        sub.pep_off = b->pep_off + lo;
        sub.n_of_mod = b->n_of_mod + lo;
        sub.max_charge = b->max_charge + lo;
        if (b->aux_off) sub.aux_off = b->aux_off + lo;
        IoReq io = {mz, inten, mk, h->io_ring[c & 1].p, h->io_ring[c & 1].p + np, h->run_stream,
                    pre_sites ? pre_sites + lo : nullptr};
        int rc = plan_create_impl(h, &sub, flags & ~(PYA_FLAG_TIMING | PYA_FLAG_KEEP), &io, pp);
        if (rc) rebase_error(h, lo);
        return rc;
    };
    typedef std::unique_ptr<pya_plan, void (*)(pya_plan *)> PlanPtr;
    pya_plan *raw = nullptr;
    int rc = make_plan(0, &raw);
    if (rc) return finish(rc);
    PlanPtr cur(raw, pya_plan_destroy), next(nullptr, pya_plan_destroy);
    for (size_t c = 0; c < nchunk; c++) {
        const uint64_t lo = cuts[c], n = cuts[c + 1] - lo;
        pya_plan *p = cur.get();
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return uploaded > c; });
            if (up_err != hipSuccess) {
                lk.unlock();
                return finish(h->hip_fail(up_err, "spectrum upload"));
            }
        }
        pya_results d_out = {mk, p->d_best_score.p, p->d_best_sig.p, p->d_n_sig_out.p, p->d_ascores.p, p->d_alt.p};
        rc = pya_plan_run(p, p->d_mz.p, p->d_inten.p, h->run_stream, &d_out);
        if (rc) return finish(rc);
        /* status + results are adjacent in the arena: one asynchronous copy into pinned memory */
        void *&pin = h->pinned_stage[c & 1];
        if (h->pinned_bytes[c & 1] < p->d2h_bytes) {
            if (pin) (void)hipHostFree(pin);
            pin = nullptr;
            h->pinned_bytes[c & 1] = 0;
            hipError_t e0 = hipHostMalloc(&pin, p->d2h_bytes + p->d2h_bytes / 4, hipHostMallocDefault);
            if (e0 != hipSuccess) return finish(h->hip_fail(e0, "pinned result buffer"));
            h->pinned_bytes[c & 1] = p->d2h_bytes + p->d2h_bytes / 4;
        }
        unsigned char *sg = (unsigned char *)pin;
        hipError_t e = hipMemcpyAsync(sg, p->arena.p + p->o_status, p->d2h_bytes, hipMemcpyDeviceToHost, h->run_stream);
        if (e != hipSuccess) return finish(h->hip_fail(e, "results copy"));
        hipEvent_t done = nullptr;                                /* chunk c finished (kernels + copy) */
        e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done, h->run_stream);
        if (e != hipSuccess) return finish(h->hip_fail(e, "event"));
        int rc_next = PYA_OK;
        if (c + 1 < nchunk) {                                     /* CPU pre-pass of the next chunk meanwhile */
            raw = nullptr;
            rc_next = make_plan(c + 1, &raw);
            next.reset(raw);
        }
        e = hipEventSynchronize(done);
        (void)hipEventDestroy(done);
        p->quiesced = e == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(mu);                   /* the chunk's ring slot may be overwritten now */
            consumed = c + 1;
            cv.notify_all();
        }
        if (e != hipSuccess) return finish(h->hip_fail(e, "results copy"));
        if (skip) std::memcpy(h->last_status.data() + lo, sg, n * sizeof(int32_t));
        const std::string keep_err = h->err;                      /* make_plan(c + 1) may have set a message */
        const int64_t keep_idx = h->err_index;
        rc = check_status(h, (const int32_t *)sg, n, skip);
        if (rc) {
            rebase_error(h, lo);
            return finish(rc);
        }
        if (rc_next) {
            h->err = keep_err;
            h->err_index = keep_idx;
            return finish(rc_next);
        }
        const size_t o = p->o_status;
        std::memcpy(out->best_score + lo, sg + (p->o_best_score - o), n * sizeof(float));
        std::memcpy(out->best_sig + lo, sg + (p->o_best_sig - o), n * sizeof(uint64_t));
        std::memcpy(out->n_sig + lo, sg + (p->o_n_sig_out - o), n * sizeof(int32_t));
        std::memcpy(out->ascores + lo * mk, sg + (p->o_ascores - o), n * mk * sizeof(float));
        std::memcpy(out->alt_mask + lo * mk, sg + (p->o_alt - o), n * mk * sizeof(uint64_t));
        cur = std::move(next);
    }
    return finish(PYA_OK);
}

int pya_score_batch(pya_handle *h, const pya_batch *b, const double *mz, const double *inten, uint32_t flags,
                    const pya_results *out) {
    if (!h || !b || !out) return PYA_ERR_ARG;
    h->last_status.clear();
    if (b->n_psm == 0) return PYA_OK;
    if (!mz || !inten) return h->fail(PYA_ERR_ARG, -1, "NULL spectrum arrays");
    if (!b->peak_off || !b->pep || !b->pep_off || !b->n_of_mod || !b->max_charge)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in batch");
    if (!out->best_score || !out->best_sig || !out->n_sig || !out->ascores || !out->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in results");
    if (b->peak_off[b->n_psm] < b->peak_off[0]) return h->fail(PYA_ERR_ARG, -1, "peak_off is not monotone");
    /* (not while the records of a pya_score_one PSM are retained in the one-PSM workspace: this call would overwrite
     * what pya_get_pep_scores / pya_calculate_ambiguity still read there) */
    const bool one_view_live = h->kept && h->kept == h->one.view;
    if (b->n_psm == 1 && !(flags & (PYA_FLAG_SKIP_INVALID | PYA_FLAG_TIMING)) && !one_view_live) {
        /* a batch of one is PyAscore.score: the low-latency path (it declines what it has no room for) */
        const bool has_aux1 = b->aux_off && b->aux_pos && b->aux_mass;
        const int64_t a0 = has_aux1 ? b->aux_off[0] : 0, a1 = has_aux1 ? b->aux_off[1] : 0;
        const int64_t P1 = b->peak_off[1] - b->peak_off[0], L1 = b->pep_off[1] - b->pep_off[0];
        if (a1 >= a0 && P1 >= 0 && L1 >= 0) {
            int rc1 = pya_score_one(h, mz + b->peak_off[0], inten + b->peak_off[0], (uint64_t)P1, b->pep + b->pep_off[0], (uint64_t)L1,
                                    b->n_of_mod[0], b->max_charge[0], has_aux1 ? b->aux_pos + a0 : nullptr,
                                    has_aux1 ? b->aux_mass + a0 : nullptr, (uint64_t)(a1 - a0), flags & PYA_FLAG_KEEP, out);
            /* the one-PSM staging now holds THIS PSM: pya_rescore_last_keep must not replay it as the caller's last
             * pya_score_one PSM (it fails with PYA_ERR_STATE instead) */
            if (!(flags & PYA_FLAG_KEEP)) h->one.have_last = false;
            if (rc1 != PYA_ERR_STATE || !h->err.empty()) return rc1;
        }
    }
    {
        /* Chunking: needed when the call does not fit the device budget, worthwhile (pipelining)
         * when there is enough PCIe traffic to hide the kernels under.  A retained batch
         * (PYA_FLAG_KEEP) stays one plan: its records are queried by PSM afterwards. */
        const size_t io_total = (size_t)(b->peak_off[b->n_psm] - b->peak_off[0]) * 16;
        if (!(flags & PYA_FLAG_KEEP) && io_total >= kChunkMin && !h->kn.no_chunks) {
            const size_t budget = workspace_budget(h);
            const ChunkCost cost = chunk_costs(h, b, out->max_k);
            double io_target = (double)kChunkTarget;
            if (h->kn.chunk_mb > 0.) io_target = h->kn.chunk_mb * 1048576.0;
            std::vector<uint64_t> cuts{0};
            double io = 0, arena = 0;
            for (uint64_t i = 0; i < b->n_psm; i++) {
                const double io2 = io + cost.io[i], ar2 = arena + cost.arena[i];
                if (i > cuts.back() && (io2 > io_target || 2.0 * io2 + ar2 > (double)budget)) {
                    cuts.push_back(i);
                    io = cost.io[i];
                    arena = cost.arena[i];
                } else {
                    io = io2;
                    arena = ar2;
                }
            }
            cuts.push_back(b->n_psm);
            if (cuts.size() > 2) return score_batch_chunked(h, b, mz, inten, flags, out, cuts, cost.sites.data());
        }
    }
    const bool host_timing = h->kn.host_timing;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!host_timing) return;
        (void)hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pya host] %-14s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    pya_plan *p = nullptr;
    IoReq io = {mz, inten, out->max_k, nullptr, nullptr, nullptr, nullptr};
    /* Big batches: the spectra (16 bytes per peak, PCIe-bound) go up on a helper thread while this
     * one runs the host pre-pass of the plan; small ones ride in the plan's single staged copy. */
    const int64_t peaks_lo = b->peak_off[0], n_peaks = b->peak_off[b->n_psm] - peaks_lo;
    std::thread uploader;
    hipError_t up_err = hipSuccess;
    if (n_peaks > 0 && (size_t)n_peaks * 16 > kStageLimit && !h->kn.no_upload_thread) {
        HIPCHK(h, hipSetDevice(h->device));
        if (h->io_buf.n < (size_t)n_peaks * 2) HIPCHK(h, h->io_buf.alloc((size_t)n_peaks * 2));
        io.d_mz_ext = h->io_buf.p;
        io.d_inten_ext = h->io_buf.p + n_peaks;
        const int device = h->device;
        uploader = std::thread([&, device]() {
            up_err = hipSetDevice(device);
            if (up_err == hipSuccess)
                up_err = hipMemcpy(io.d_mz_ext, mz + peaks_lo, (size_t)n_peaks * 8, hipMemcpyHostToDevice);
            if (up_err == hipSuccess)
                up_err = hipMemcpy(io.d_inten_ext, inten + peaks_lo, (size_t)n_peaks * 8, hipMemcpyHostToDevice);
        });
    }
    int rc = plan_create_impl(h, b, flags & ~PYA_FLAG_TIMING, &io, &p);
    if (uploader.joinable()) uploader.join();
    if (rc) return rc;
    if (up_err != hipSuccess) {
        pya_plan_destroy(p);
        return h->hip_fail(up_err, "spectrum upload");
    }
    std::unique_ptr<pya_plan, void (*)(pya_plan *)> guard(p, pya_plan_destroy);
    lap("plan + h2d");
    const uint64_t n = b->n_psm;
    const uint32_t mk = out->max_k;
    pya_results d_out = {mk, p->d_best_score.p, p->d_best_sig.p, p->d_n_sig_out.p, p->d_ascores.p, p->d_alt.p};
    rc = pya_plan_run(p, p->d_mz.p, p->d_inten.p, nullptr, &d_out);
    if (rc) return rc;
    if (p->d2h_bytes <= kStageLimit) {
        /* status and results are adjacent in the arena: one copy, which also waits for the kernels */
        h->stage.resize(std::max(h->stage.size(), p->d2h_bytes));
        unsigned char *sg = h->stage.data();
        HIPCHK(h, hipMemcpy(sg, p->arena.p + p->o_status, p->d2h_bytes, hipMemcpyDeviceToHost));
        lap("kernels + d2h");
        const bool skip = (flags & PYA_FLAG_SKIP_INVALID) != 0;
        if (skip) h->last_status.assign((const int32_t *)sg, (const int32_t *)sg + n);
        rc = check_status(h, (const int32_t *)sg, n, skip);
        if (rc) return rc;
        const size_t o = p->o_status;
        std::memcpy(out->best_score, sg + (p->o_best_score - o), n * sizeof(float));
        std::memcpy(out->best_sig, sg + (p->o_best_sig - o), n * sizeof(uint64_t));
        std::memcpy(out->n_sig, sg + (p->o_n_sig_out - o), n * sizeof(int32_t));
        std::memcpy(out->ascores, sg + (p->o_ascores - o), n * mk * sizeof(float));
        std::memcpy(out->alt_mask, sg + (p->o_alt - o), n * mk * sizeof(uint64_t));
    } else {
        rc = pya_plan_check(p);
        if (rc) return rc;
        lap("kernels");
        HIPCHK(h, hipMemcpy(out->best_score, p->d_best_score.p, n * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->best_sig, p->d_best_sig.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->n_sig, p->d_n_sig_out.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->ascores, p->d_ascores.p, n * mk * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(out->alt_mask, p->d_alt.p, n * mk * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    lap("d2h");
    if (flags & PYA_FLAG_KEEP) {
        if (h->kept) pya_plan_destroy(h->kept);
        h->kept = guard.release();
    }
    return PYA_OK;
}

} /* extern "C" */

/* ------------------------------------------------------------------------------------------------------ */
/* pya_score_one: one PSM per call, lowest latency (tiny_batch.hip: pya_one_kernel)                        */
/* ------------------------------------------------------------------------------------------------------ */
namespace {
/* layout of the pinned block: [flag 64 B | status 64 B | best_score, n_sig, best_sig 64 B | ascores 64 x 4 |
 * alt masks 64 x 8 | m/z PYA_FAST_PEAKS x 8 | intensity PYA_FAST_PEAKS x 8] */
const size_t kOneFlag = 0, kOneStatus = 64, kOneBest = 128, kOneAsc = 192, kOneAlt = 448, kOneMz = 1024,
             kOneInt = kOneMz + (size_t)PYA_FAST_PEAKS * 8, kOneBytes = kOneInt + (size_t)PYA_FAST_PEAKS * 8;

int one_prepare(pya_handle *h, uint32_t n_sig) {
    pya_handle::One &o = h->one;
    if (!o.host) {
        HIPCHK(h, hipHostMalloc((void **)&o.host, kOneBytes, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(h, hipHostGetDevicePointer((void **)&o.host_dev, o.host, 0));
        std::memset(o.host, 0, 1024);
        HIPCHK(h, hipStreamCreateWithFlags(&o.stream, hipStreamNonBlocking));
    }
    if (!o.ws.p || o.sig_cap < n_sig) {
        /* device workspace of one PSM: small arrays, the retained table, grid, per-signature scores / records / order */
        const uint32_t cap = std::max<uint32_t>(1024, next_pow2(n_sig));
        const size_t bytes = 4096 + ((size_t)PYA_FAST_PEAKS + 8) * sizeof(PeakEntry) + PYA_GRID_CELLS * 2 + 4096 +
                             (size_t)cap * (4 + PYA_REC_WORDS * 4 + 4) + 1024;
        HIPCHK(h, hipStreamSynchronize(o.stream));
        /* the retained view (if any) points into the allocation that goes away */
        if (o.view && h->kept == o.view) h->kept = nullptr;
        o.last_keep = false;
        HIPCHK(h, o.ws.alloc(bytes));
        HIPCHK(h, hipMemset(o.ws.p, 0, bytes));
        o.sig_cap = cap;
        unsigned char *w = o.ws.p;
        BatchDev &d = o.dev;
        std::memset(&d, 0, sizeof d);
        size_t at = 0;
        auto take = [&](size_t n) { unsigned char *q = w + at; at += (n + 255) & ~(size_t)255; return q; };
        d.peak_off = (const int64_t *)take(16);
        d.pep_off = (const int64_t *)take(16);
        d.aux_off = (const int64_t *)take(16);
        d.sig_off = (const int64_t *)take(16);
        d.ret_off = (const int64_t *)take(16);
        d.pep = (const uint8_t *)take(PYA_MAX_L);
        d.n_of_mod = (const int32_t *)take(4);
        d.max_charge = (const int32_t *)take(4);
        d.aux_pos = (const uint32_t *)take(PYA_ONE_MAX_AUX * 4);
        d.aux_mass = (const float *)take(PYA_ONE_MAX_AUX * 4);
        d.n_sites = (const uint8_t *)take(4);
        d.n_sig = (const uint32_t *)take(4);
        d.order_off = (const uint32_t *)take(4);
        d.desc = (const uint64_t *)take(PYA_DESC_WORDS * 8);
        d.status = (int32_t *)take(4);
        d.ret_n = (uint32_t *)take(4);
        d.ws_top = (uint32_t *)take(16);
        uint32_t *redo = (uint32_t *)take(4 * 80);
        d.redo_count = redo;
        d.redo_ids = redo + 64;
        d.redo3_count = redo + 66;
        d.redo3_ids = redo + 70;
        d.redo3b_count = redo + 67;
        d.redo3b_ids = redo + 72;
        d.redo4_count = redo + 68;
        d.redo4_ids = redo + 74;
        d.grid = (uint16_t *)take(PYA_GRID_CELLS * 2);
        d.ret = (PeakEntry *)take(((size_t)PYA_FAST_PEAKS + 8) * sizeof(PeakEntry));
        d.ws = (float *)take((size_t)cap * 4);
        d.rec = (uint32_t *)take((size_t)cap * PYA_REC_WORDS * 4);
        d.sorted_idx = (uint32_t *)take((size_t)cap * 4);
        d.mz = (const double *)(o.host_dev + kOneMz);
        d.inten = (const double *)(o.host_dev + kOneInt);
        d.best_score = (float *)(o.host_dev + kOneBest);
        d.n_sig_out = (int32_t *)(o.host_dev + kOneBest + 8);
        d.best_sig = (uint64_t *)(o.host_dev + kOneBest + 16);
        d.ascores = (float *)(o.host_dev + kOneAsc);
        d.alt_mask = (uint64_t *)(o.host_dev + kOneAlt);
    }
    return PYA_OK;
}

/* launches the kernel for o.meta (the spectrum is in the pinned block) and waits for its results */
int one_run(pya_handle *h, bool keep, uint32_t max_k) {
    pya_handle::One &o = h->one;
    const OneMeta &m = o.meta;
    BatchDev d = o.dev;
    d.order_tab = h->d_order.p;
    d.inv_tab = h->d_inv.p;
    d.binom = h->d_binom.p;
    d.cfg = h->d_cfg.p;
    d.lut = h->d_lut.p;
    d.lut_off = h->d_lut_off.p;
    d.lut_n_max = h->lut_uploaded_n - 1;
    d.max_k = max_k;
    d.keep = keep ? 1u : 0u;
    d.debug = h->kn.debug & 0xffffu;
    /* caps of this one PSM (the rules of the plan's buckets, for a bucket of one) */
    Bucket bk;
    const uint32_t L = m.L, z = (uint32_t)m.max_charge, k = (uint32_t)m.n_of_mod, ns = m.n_sites, N = m.n_sig;
    const uint32_t n_uniq = (uint32_t)h->cfg.n_uniq, n_types = (uint32_t)h->cfg.n_types;
    const uint32_t per_type = (L - 1) * z * n_uniq;
    bk.n_cap = N;
    bk.list_cap = next_pow2(std::max<uint32_t>(per_type, 1));
    bk.pos_cap = std::max<uint32_t>(L - 1, 1);
    bk.n_types = n_types;
    bk.k_max = std::max<uint32_t>(k, 1);
    bk.push_max = std::max<uint32_t>(k < ns ? k * (ns - k) : 1, 1);
    bk.z_max = z;
    const uint32_t cap = (m.n_peaks + 31u) & ~31u;
    const uint32_t prefix = (N >= 128 && !h->kn.no_prefix) ? 1u : 0u;
    const bool plain_types = h->cfg.n_nl == 0 && h->cfg.n_fwd <= 1 && h->cfg.n_types - h->cfg.n_fwd <= 1;
    const uint32_t compact = (plain_types && z == 1) ? 1u : 0u;
    const bool both = h->cfg.n_fwd > 0 && h->cfg.n_fwd < h->cfg.n_types;
    uint32_t use_fused = 0, f_n_cap = 4, f_stride = 8, f_ent = 1, f_push = 8;
    const uint32_t frags = (both ? 2u : 1u) * (L - 1) * z;
    if (!keep && plain_types && !h->kn.no_fused && k < ns && N > 0 && N <= (both ? 32u : 64u) && frags <= 255u) {
        use_fused = both ? 1u : 2u;
        f_n_cap = (N + 3u) & ~3u;
        f_stride = (both ? 2u : 1u) * f_n_cap + 4u;
        f_ent = std::max<uint32_t>((L - 1) * z, 1);
        f_push = std::max<uint32_t>(8u, bk.push_cap());
    }
    if (pya_one_lds_bytes(cap, prefix, h->cfg.n_nl != 0, compact, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(), bk.sb(), use_fused,
                          f_n_cap, f_stride, f_ent, f_push, z > 1) > kMaxLds)
        return h->fail(PYA_ERR_LIMIT, 0, "LDS budget exceeded for this PSM");
    volatile uint32_t *flag = (volatile uint32_t *)(o.host + kOneFlag);
    int e = pya_launch_one(&d, &m, cap, prefix, h->cfg.n_nl != 0 ? 1u : 0u, compact, bk.push_cap(), bk.n_cap, bk.pos_cap, bk.pool_cap(),
                           bk.sb(), bk.gtp(), use_fused, f_n_cap, f_stride, f_ent, f_push, z > 1 ? 1u : 0u,
                           (int32_t *)(o.host_dev + kOneStatus), (uint32_t *)(o.host_dev + kOneFlag), o.stream);
    if (e) return h->hip_fail((hipError_t)e, "score (one PSM) launch");
    /* the kernel's last store is the sequence number: poll it (a stream synchronisation costs several
     * microseconds more); give up after two seconds and ask the runtime what happened */
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *flag != m.seq; spins++) {
        __builtin_ia32_pause();
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
            HIPCHK(h, hipStreamSynchronize(o.stream));
            if (*flag != m.seq) return h->fail(PYA_ERR_HIP, 0, "the kernel finished without publishing its results");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    o.last_keep = keep;
    o.last_max_k = max_k;
    /* a retained view for pya_get_pep_scores / pya_calculate_ambiguity */
    if (keep) {
        if (!o.view) o.view = new pya_plan;
        pya_plan *p = o.view;
        p->h = h;
        p->flags = PYA_FLAG_KEEP;
        p->n_psm = 1;
        p->peak_cap = cap;
        p->max_k = max_k;
        p->sig_off = {0, (int64_t)N};
        p->pep_off = {0, (int64_t)L};
        p->n_sig = {N};
        p->order_off = {m.order_off};
        p->max_charge = {m.max_charge};
        p->d_rec.adopt(d.rec, (size_t)N * PYA_REC_WORDS);
        p->d_sorted.adopt(d.sorted_idx, N);
        p->d_ws.adopt(d.ws, N);
        p->dev = d;
        p->quiesced = true;
        p->ran = true;
        p->last_stream = o.stream;
        if (h->kept && h->kept != p) pya_plan_destroy(h->kept);
        h->kept = p;
    } else if (h->kept && h->kept == o.view) {
        h->kept = nullptr;                                  /* the view described the previous PSM */
    }
    return PYA_OK;
}
}  // namespace

extern "C" int pya_score_one(pya_handle *h, const double *mz, const double *inten, uint64_t n_peaks, const uint8_t *pep,
                             uint64_t L, int32_t n_of_mod, int32_t max_charge, const uint32_t *aux_pos, const float *aux_mass,
                             uint64_t n_aux, uint32_t flags, const pya_results *out) {
    if (!h || !out) return PYA_ERR_ARG;
    h->err.clear();
    h->err_index = -1;
    h->last_status.clear();
    if (h->kn.no_tiny) return PYA_ERR_STATE;               /* (route switch of the tests: the kernel-per-stage path) */
    if (!mz || !inten || !pep) return h->fail(PYA_ERR_ARG, -1, "NULL array");
    if (!out->best_score || !out->best_sig || !out->n_sig || !out->ascores || !out->alt_mask)
        return h->fail(PYA_ERR_ARG, -1, "NULL array in results");
    if (n_aux && (!aux_pos || !aux_mass)) return h->fail(PYA_ERR_ARG, -1, "NULL fixed-modification arrays");
    /* validation: what plan_create_impl checks for a PSM (same messages) */
    if (n_peaks == 0) return h->fail(PYA_ERR_PSM, 0, "PSM 0: empty spectrum");
    if (n_peaks > PYA_MAX_PEAKS) return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %llu peaks exceed the limit of %d", (unsigned long long)n_peaks, PYA_MAX_PEAKS);
    if (L < 1 || L > PYA_MAX_PEPTIDE_LEN)
        return h->fail(L < 1 ? PYA_ERR_PSM : PYA_ERR_LIMIT, 0, "PSM 0: peptide length %lld outside 1..%d", (long long)L, PYA_MAX_PEPTIDE_LEN);
    if (n_of_mod < 0) return h->fail(PYA_ERR_PSM, 0, "PSM 0: negative n_of_mod");
    if (max_charge < 1 || max_charge > 16) return h->fail(PYA_ERR_PSM, 0, "PSM 0: max_fragment_charge %d outside 1..16", max_charge);
    uint32_t ns = 0;
    for (uint64_t j = 0; j < L; j++) {
        if (!h->is_residue[pep[j]])
            return h->fail(PYA_ERR_PSM, 0, "PSM 0: unknown residue '%c' at position %lld", (char)pep[j], (long long)(j + 1));
        if (h->letter_modifiable((char)pep[j], (size_t)j, (size_t)L)) ns++;
    }
    for (uint64_t a = 0; a < n_aux; a++)
        if (aux_pos[a] > (uint32_t)L) return h->fail(PYA_ERR_PSM, 0, "PSM 0: aux_mod_pos %u beyond the peptide", aux_pos[a]);
    if (ns > PYA_MAX_SITES) return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %u modifiable residues exceed %d", ns, PYA_MAX_SITES);
    uint64_t N = 0;
    if ((uint32_t)n_of_mod <= ns) {
        uint64_t &cached = h->binom_cache[ns][n_of_mod];
        if (cached == 0) cached = binom(ns, (uint32_t)n_of_mod);
        N = cached;
    }
    if (N > PYA_MAX_SIGNATURES)
        return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: C(%u,%d) site assignments exceed the limit of %d", ns, n_of_mod, PYA_MAX_SIGNATURES);
    const uint32_t per_type = (uint32_t)(L - 1) * (uint32_t)max_charge * (uint32_t)h->cfg.n_uniq;
    if (per_type > PYA_MAX_FRAGMENTS_PER_TYPE)
        return h->fail(PYA_ERR_LIMIT, 0, "PSM 0: %u fragments per ion type exceed %d", per_type, PYA_MAX_FRAGMENTS_PER_TYPE);
    /* (beyond a limit of the fast kernels: the caller takes the batch path, which has the general kernel) */
    if (h->n_top != PYA_NTOP || n_peaks > PYA_FAST_PEAKS || L > PYA_FAST_PEPTIDE_LEN || N > PYA_FAST_SIGNATURES || per_type > PYA_FAST_FRAGMENTS_PER_TYPE) return PYA_ERR_STATE;
    if (n_aux > PYA_ONE_MAX_AUX || (uint32_t)n_of_mod > 64u) return PYA_ERR_STATE;     /* (the caller takes the batch path) */
    if (out->max_k < (uint32_t)std::max(n_of_mod, 1)) return h->fail(PYA_ERR_ARG, -1, "results.max_k is smaller than n_of_mod");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = sync_config(h);
    if (rc) return rc;
    uint32_t ooff = 0;
    if (N) {
        uint32_t &co = h->shape_cache[ns][n_of_mod];
        if (co == 0xffffffffu) co = shape_offset(h, ns, (uint32_t)n_of_mod);
        ooff = co;
    }
    rc = ensure_lut(h, per_type * (uint32_t)h->cfg.n_types);
    if (rc) return rc;
    if (h->order_uploaded != h->order_tab.size() || !h->d_order.p) {
        HIPCHK(h, h->d_order.upload(h->order_tab.data(), h->order_tab.size()));
        HIPCHK(h, h->d_inv.upload(h->inv_tab.data(), h->inv_tab.size()));
        if (!h->d_binom.p) {
            std::vector<uint32_t> bt(64 * 64);
            for (uint32_t pp = 0; pp < 64; pp++)
                for (uint32_t t = 0; t < 64; t++) bt[pp * 64 + t] = (uint32_t)std::min<uint64_t>(binom(pp, t), 0xffffffffull);
            HIPCHK(h, h->d_binom.upload(bt.data(), bt.size()));
        }
        HIPCHK(h, hipDeviceSynchronize());
        h->order_uploaded = h->order_tab.size();
    }
    rc = one_prepare(h, (uint32_t)N);
    if (rc) return rc;
    pya_handle::One &o = h->one;
    /* inputs: the spectrum into the pinned block, everything else into the kernel's arguments */
    std::memcpy(o.host + kOneMz, mz, (size_t)n_peaks * 8);
    std::memcpy(o.host + kOneInt, inten, (size_t)n_peaks * 8);
    OneMeta &m = o.meta;
    std::memset(&m, 0, sizeof m);
    std::memcpy(m.pep, pep, (size_t)L);
    m.n_peaks = (uint32_t)n_peaks;
    m.L = (uint32_t)L;
    m.n_aux = (uint32_t)n_aux;
    m.n_sig = (uint32_t)N;
    m.order_off = ooff;
    m.seq = ++o.seq ? o.seq : ++o.seq;
    m.n_of_mod = n_of_mod;
    m.max_charge = max_charge;
    m.n_sites = ns;
    for (uint64_t a = 0; a < n_aux; a++) {
        m.aux_pos[a] = aux_pos[a];
        m.aux_mass[a] = aux_mass[a];
    }
    m.desc[0] = 0;
    m.desc[1] = 0;
    m.desc[2] = 0;
    m.desc[3] = 0;
    m.desc[4] = (uint64_t)L | (uint64_t)n_aux << 16 | (uint64_t)((uint32_t)n_of_mod & 0xffffu) << 32 | (uint64_t)ns << 48 |
                (uint64_t)((uint32_t)max_charge & 0xffu) << 56;
    m.desc[5] = (uint64_t)N | (uint64_t)ooff << 32;
    o.have_last = true;
    const uint32_t mk = out->max_k;
    if (mk > 64) return PYA_ERR_STATE;
    rc = one_run(h, (flags & PYA_FLAG_KEEP) != 0, mk);
    if (rc) return rc;
    const int32_t st = *(const int32_t *)(o.host + kOneStatus);
    rc = check_status(h, &st, 1, false);
    if (rc) return rc;
    out->best_score[0] = *(const float *)(o.host + kOneBest);
    out->n_sig[0] = *(const int32_t *)(o.host + kOneBest + 8);
    out->best_sig[0] = *(const uint64_t *)(o.host + kOneBest + 16);
    std::memcpy(out->ascores, o.host + kOneAsc, (size_t)mk * 4);
    std::memcpy(out->alt_mask, o.host + kOneAlt, (size_t)mk * 8);
    return PYA_OK;
}

/* the last pya_score_one PSM once more, retained (PYA_FLAG_KEEP): what PyAscore.pep_scores and
 * calculate_ambiguity need; the spectrum and the scalars are still where the last call put them */
extern "C" int pya_rescore_last_keep(pya_handle *h) {
    if (!h) return PYA_ERR_ARG;
    if (!h->one.have_last) return h->fail(PYA_ERR_STATE, -1, "no PSM scored with pya_score_one yet");
    if (h->one.last_keep && h->kept == h->one.view) return PYA_OK;
    HIPCHK(h, hipSetDevice(h->device));
    h->one.meta.seq = ++h->one.seq ? h->one.seq : ++h->one.seq;
    return one_run(h, true, h->one.last_max_k);
}

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
    if (L > PYA_FAST_PEPTIDE_LEN || h->n_top != PYA_NTOP)
        return h->fail(PYA_ERR_LIMIT, (int64_t)psm, "calculate_ambiguity takes peptides of up to %d residues and n_top = %d (everything "
                       "else is scored by the general kernel only: the Ascores are in the results)", PYA_FAST_PEPTIDE_LEN, PYA_NTOP);
    const uint32_t list_cap = next_pow2(std::max<uint32_t>(1, (uint32_t)(L - 1) * (uint32_t)p->max_charge[psm] *
                                                                  (uint32_t)h->cfg.n_uniq));
    float host_scores[2 * PYA_NTOP];
    std::memcpy(host_scores, ref_scores, PYA_NTOP * sizeof(float));
    std::memcpy(host_scores + PYA_NTOP, other_scores, PYA_NTOP * sizeof(float));
    DevBuf<float> d_scores, d_out;
    refresh_shared(p);
    HIPCHK(h, d_scores.upload(host_scores, 2 * PYA_NTOP));
    HIPCHK(h, d_out.alloc(2));
    int e = pya_launch_ambiguity(&p->dev, (uint32_t)psm, p->peak_cap, list_cap, ref_bits, other_bits, d_scores.p,
                                 ref_ws, other_ws, d_out.p, nullptr);
    if (e) return h->hip_fail((hipError_t)e, "ambiguity launch");
    float res[2];
    HIPCHK(h, hipMemcpy(res, d_out.p, sizeof res, hipMemcpyDeviceToHost));
    if (res[1] != 0.f) return h->fail(PYA_ERR_LIMIT, (int64_t)psm, "trial count outside the score table");
    *out = res[0];
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
