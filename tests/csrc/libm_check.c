/* Test helper (CPU): holds slidingwindowdecoder_amd/csrc/swd_libm.h to the host C library, bit for bit.
 * Built by tests/test_libm_restatement.py with gcc; returns the number of arguments where the results differ. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "swd_libm.h"

static uint64_t rng_state;
static double urand(void) { /* splitmix64 -> [0, 1) */
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1p-53;
}
static int same(double a, double b) {
    uint64_t x, y;
    memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    return x == y || (a != a && b != b);
}

long swd_check_exp(long n, uint64_t seed, double *first_bad) {
    long bad = 0;
    rng_state = seed;
    for (long i = 0; i < n; ++i) {
        double x;
        switch (i % 6) {
        case 0: x = (urand() - 0.5) * 100.0; break;
        case 1: x = (urand() - 0.5) * 2.0; break;
        case 2: x = -urand() * 760.0; break;          /* down to the subnormal results and underflow */
        case 3: x = urand() * 712.0; break;           /* up to overflow */
        case 4: x = (urand() - 0.5) * 1e-3; break;
        default: x = (urand() - 0.5) * exp((urand() - 0.5) * 100.0); break;
        }
        if (!same(exp(x), swd_exp(x))) { if (!bad && first_bad) *first_bad = x; ++bad; }
    }
    const double special[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 1e308, -1e308, 709.782712893384, -745.2, 0x1p-60, 512.0, -512.0, 1023.9, -1023.9};
    for (unsigned i = 0; i < sizeof special / sizeof special[0]; ++i)
        if (!same(exp(special[i]), swd_exp(special[i]))) { if (!bad && first_bad) *first_bad = special[i]; ++bad; }
    return bad;
}

long swd_check_log1p(long n, uint64_t seed, double *first_bad) {
    long bad = 0;
    rng_state = seed;
    for (long i = 0; i < n; ++i) {
        double x;
        switch (i % 6) {
        case 0: x = urand() * 2.0 - 0.999; break;
        case 1: x = exp((urand() - 0.5) * 1400.0); break;
        case 2: x = -exp(-urand() * 700.0); break;
        case 3: x = (urand() - 0.5) * 1e-6; break;
        case 4: x = exp((urand() - 0.5) * 80.0); break;
        default: x = urand() * 100.0; break;
        }
        if (!same(log1p(x), swd_log1p(x))) { if (!bad && first_bad) *first_bad = x; ++bad; }
    }
    const double special[] = {0.0, -0.0, -1.0, -2.0, INFINITY, NAN, 1e308, 0x1p-60, 0x1p-30, 0x1p53, 0x1p52, -0.2929, 0.41422, 1.0};
    for (unsigned i = 0; i < sizeof special / sizeof special[0]; ++i)
        if (!same(log1p(special[i]), swd_log1p(special[i]))) { if (!bad && first_bad) *first_bad = special[i]; ++bad; }
    return bad;
}
