/* score_table.cpp -- host-built table of the per-depth binomial scores.
 *
 * score[d][n][k] = | -10 * log10 P(X >= k) |,  X ~ Binomial(n, p_d),  p_d = 2*mz_error*d/100
 *
 * Replaces BinomialDist / LogMath (cpp/Util.cpp:16-83) and the score expression of
 * Ascore::calculateFullScores / calculateAmbiguity (cpp/Ascore.cpp:33, :127-133, :200-207).
 * The reference evaluates this chain in float32 with libm logf/expf/log, re-rounding after every
 * step; the result is a pure function of (d, n, k) and is far (1e-4) from the float64 value, so
 * it is tabulated here with the identical operation order and read by the kernels.  This file is
 * compiled by g++ with -ffp-contract=off (the reference's own compiler and flags class) so the
 * float chain is reproduced step for step; it is configuration data, not a CPU scoring path.
 */
#include <cmath>
#include <cstdint>
#include <vector>

#include "binom_chain.h"

using pya_chain::log_sum;

/* Appends rows n = off.size() .. n_to (inclusive) for the n_top depths.  Row layout:
 * lut[off[n] + d*(n+1) + k].  off gets one entry per n. */
void pya_score_table_extend(float mz_error, uint32_t n_top, uint32_t n_to, std::vector<float> &lut,
                            std::vector<uint32_t> &off) {
    std::vector<double> logd(n_to + 2, 0.);
    for (uint32_t m = 1; m <= n_to + 1; m++) logd[m] = std::log((double)m);
    std::vector<float> log_p(n_top), log_q(n_top);
    for (uint32_t d = 1; d <= n_top; d++) {
        float t = (2 * mz_error) * (float)d;                         /* Ascore.cpp:33 */
        float p = (float)((double)t / 100.);
        log_p[d - 1] = std::log(p);                                  /* logf */
        log_q[d - 1] = (float)std::log(1. - (double)p);
    }
    const double log10e = std::log10(std::exp(1.0));
    std::vector<float> coef, tail;
    for (uint32_t n = (uint32_t)off.size(); n <= n_to; n++) {
        coef.assign(n + 1, 0.f);
        for (uint32_t k = 0; k <= n; k++) {                          /* Util.cpp:28-41 */
            uint32_t kk = (n - k) < k ? (n - k) : k;
            float c = 0.f;
            for (uint32_t m = n - kk + 1; m <= n; m++) c = (float)((double)c + logd[m]);
            for (uint32_t m = 2; m <= kk; m++) c = (float)((double)c - logd[m]);
            coef[k] = c;
        }
        uint32_t base = (uint32_t)lut.size();
        off.push_back(base);
        lut.resize((size_t)base + (size_t)n_top * (n + 1));
        tail.assign(n + 2, 0.f);
        for (uint32_t d = 0; d < n_top; d++) {
            tail[n + 1] = -INFINITY;
            for (uint32_t j = n; j >= 1; j--) {                      /* Util.cpp:52-79 */
                float pmf = (coef[j] + (float)j * log_p[d]) + (float)(n - j) * log_q[d];
                tail[j] = log_sum(tail[j + 1], pmf);
            }
            tail[0] = 0.f;
            float *row = lut.data() + base + (size_t)d * (n + 1);
            for (uint32_t k = 0; k <= n; k++) {
                float l10 = (float)(log10e * (double)tail[k]);       /* Util.cpp:81-83 */
                row[k] = std::abs(-10 * l10);                        /* Ascore.cpp:127-133 */
            }
        }
    }
}
