/* scripts/divcheck.c -- the FMA sequence of charge_mz (csrc/device_common.hip.h) against the IEEE division on the host:
 *   gcc -O2 -mfma -ffp-contract=off scripts/divcheck.c -o /tmp/divcheck -lm && /tmp/divcheck
 * q0 = a y, two corrections q <- fma(fma(-b, q, a), y, q) with y = RN(1 / b): for every b = 3 .. 255 that is not a power of two,
 * ITERS random a per b (default 2 000 000: 494 000 000 in all).  Prints how often the result differs from a / b (never). */
#include <math.h>
#ifndef ITERS
#define ITERS 2000000
#endif
#include <stdio.h>
#include <stdint.h>
#include <string.h>
static inline uint64_t rng(uint64_t *s){ *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
int main(void){
    uint64_t s = 88172645463325252ull; unsigned long long bad = 0, n = 0, bad1 = 0;
    for (int z = 3; z <= 255; z++) {
        if ((z & (z - 1)) == 0) continue;
        const double b = (double)z, y = 1.0 / b;
        for (int it = 0; it < ITERS; it++) {
            uint64_t r = rng(&s);
            /* a = m + z * 1.007825 with m anywhere from 1 to 1e6, random mantissa */
            double a = ldexp((double)(r >> 11) * (1.0 / 9007199254740992.0) + 1.0, (int)(rng(&s) % 21));
            double q0 = a * y;
            double r0 = fma(-b, q0, a);
            double q1 = fma(r0, y, q0);
            double r1 = fma(-b, q1, a);
            double q = fma(r1, y, q1);
            double t = a / b;
            n++;
            if (memcmp(&q, &t, 8)) bad++;
            if (memcmp(&q1, &t, 8)) bad1++;
        }
    }
    printf("samples %llu  two corrections differ %llu  one correction differs %llu\n", n, bad, bad1);
    return bad != 0;
}
