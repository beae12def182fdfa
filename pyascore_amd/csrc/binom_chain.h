/* binom_chain.h -- the reference's float32 log-space binomial arithmetic, host side.
 *
 * LogMath / BinomialDist (cpp/Util.cpp:16-83) evaluate every step in float with libm and re-round
 * after each operation; the result is a pure function of (p, k, n) but far (1e-4) from the float64
 * value, so anything that has to agree with the reference repeats the chain operation by
 * operation.  Shared by the score table the kernels read (score_table.cpp) and by the scripting
 * classes PyLogMath / PyBinomialDist (aux_api.cpp).  Compiled by g++ with -ffp-contract=off.
 */
#ifndef PYA_BINOM_CHAIN_H
#define PYA_BINOM_CHAIN_H

#include <cmath>
#include <cstdint>

namespace pya_chain {

inline float log_sum(float a, float b) {                             /* Util.cpp:16-26 */
    if (std::isinf(a)) return b;
    if (std::isinf(b)) return a;
    float m = a < b ? b : a;                                         /* std::max(a, b) */
    float s = std::exp(a - m) + std::exp(b - m);
    return m + std::log(s);
}

/* log C(n, k): float accumulator, every log taken in double and added to it (Util.cpp:28-41) */
inline float log_bin_coef(uint64_t k, uint64_t n) {
    const uint64_t kk = (n - k) < k ? (n - k) : k;
    float c = 0.f;
    for (uint64_t m = n - kk + 1; m <= n; m++) c = (float)((double)c + std::log((double)m));
    for (uint64_t m = 2; m <= kk; m++) c = (float)((double)c - std::log((double)m));
    return c;
}

struct Binomial {                                                    /* Util.cpp:47-59 */
    float log_p, log_q;
    explicit Binomial(float p) : log_p(std::log(p)), log_q((float)std::log(1. - (double)p)) {}
    float log_pmf_from(float coef, uint64_t k, uint64_t n) const {
        return (coef + (float)k * log_p) + (float)(n - k) * log_q;
    }
    float log_pmf(uint64_t k, uint64_t n) const { return log_pmf_from(log_bin_coef(k, n), k, n); }
    /* log P(X >= k): log-sum of the pmf from n down to k (Util.cpp:61-79); 0 for k == 0 */
    float log_pvalue(uint64_t k, uint64_t n) const {
        if (k == 0) return 0.f;
        float tail = -INFINITY;
        for (uint64_t j = n; j >= k; j--) tail = log_sum(tail, log_pmf(j, n));
        return tail;
    }
};

inline float log10_of(float log_pvalue) {                            /* Util.cpp:81-83 */
    return (float)(std::log10(std::exp(1.0)) * (double)log_pvalue);
}

}  // namespace pya_chain
#endif
