/* TEST INFRASTRUCTURE (like everything under oracle/): pins csrc/smz_glibc_math.hpp -- the product's restatement of glibc's
   log() / pow(), the same source the device compiles -- against THIS machine's libm, argument by argument.  Only tests/ load
   the library built from this file (oracle/libglibccheck.so); the product never does.
   Built with -ffp-contract=off so that gcc fuses exactly what the header spells as __builtin_fma and nothing else (-mfma only
   makes those calls one instruction). */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../stochastic-muzero_amd/csrc/smz_glibc_math.hpp"

static uint64_t next_u64(uint64_t *s) {                 /* splitmix64 */
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static double u53(uint64_t *s) { return (double)(next_u64(s) >> 11) * (1.0 / 9007199254740992.0); }   /* numpy's random_sample grid */
static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static int same(double a, double b) { return memcmp(&a, &b, 8) == 0 || (a != a && b != b); }

static double shape_of(uint64_t *s, int mode) {
    static const double fixed[] = {0.25, 0.3, 0.03, 0.5, 1.0 / 3.0, 0.9, 0.15, 0.75, 0.01, 0.999};
    if (mode & 1) return fixed[next_u64(s) % 10];
    double a = u53(s);
    return a < 1e-3 ? 1e-3 : a;
}

void glc_log_array(const double *x, double *out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = smz_glibc_log(x[i]); }
void glc_pow_array(const double *x, const double *y, double *out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = smz_glibc_pow(x[i], y[i]); }
void glc_libm_log_array(const double *x, double *out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = log(x[i]); }
void glc_libm_pow_array(const double *x, const double *y, double *out, int64_t n) { for (int64_t i = 0; i < n; i++) out[i] = pow(x[i], y[i]); }

/* the argument a mode draws: what the gamma sampler passes, then wider nets */
static double log_arg(uint64_t *s, int mode) {
    switch (mode) {
    case 0: return 1.0 - u53(s);                                          /* legacy_standard_exponential */
    case 1: return (1.0 - u53(s)) / shape_of(s, (int)next_u64(s));        /* Y = -log((1 - U) / shape) */
    case 2: return from_bits(next_u64(s) & 0x7fefffffffffffffull);        /* any finite non-negative double, subnormals included */
    case 3: return 1.0 + (u53(s) - 0.5) * 0.15;                           /* both sides of the near-1 interval's ends */
    default: return from_bits(0x3fee000000000000ull + (next_u64(s) % 0x3100000000000ull) - 0x4000000000ull);
    }
}
int64_t glc_check_log(uint64_t seed, int64_t n, int mode, double *bad) {
    int64_t wrong = 0;
    for (int64_t i = 0; i < n; i++) {
        const double x = log_arg(&seed, mode);
        if (!same(smz_glibc_log(x), log(x))) { if (!wrong && bad) bad[0] = x; wrong++; }
    }
    return wrong;
}

int64_t glc_check_pow(uint64_t seed, int64_t n, int mode, double *bad) {
    int64_t wrong = 0;
    for (int64_t i = 0; i < n; i++) {
        double x, y;
        if (mode == 0) {                                                  /* X = pow(U, 1 / shape) */
            const double shape = shape_of(&seed, (int)next_u64(&seed));
            x = u53(&seed) * (1.0 - shape); y = 1.0 / shape;
        } else if (mode == 1) {                                           /* X = pow(1 - shape + shape Y, 1 / shape) */
            const double shape = shape_of(&seed, (int)next_u64(&seed));
            const double U = 1.0 - shape + u53(&seed) * shape;
            const double Y = -log((1.0 - U) / shape);
            x = 1.0 - shape + shape * Y; y = 1.0 / shape;
        } else if (mode == 2) {                                           /* any positive finite x, y over 2^-70 .. 2^70 */
            x = from_bits(next_u64(&seed) & 0x7fefffffffffffffull);
            y = from_bits(((0x3ffull - 70 + next_u64(&seed) % 141) << 52) | (next_u64(&seed) >> 12));
        } else if (mode == 3) {                                           /* results around the under- / overflow thresholds */
            x = from_bits(next_u64(&seed) & 0x7fefffffffffffffull);
            const double lx = log(x);
            const double target = (next_u64(&seed) & 1 ? -1.0 : 1.0) * (690.0 + 70.0 * u53(&seed));
            y = lx != 0.0 ? target / lx : 1.0;
            if (!(y > 0.0) || y > 1e300) y = 3.5;
        } else {                                                          /* the policy temperature: p ^ (1 / T) */
            x = u53(&seed); y = 1.0 / (0.3 + 0.7 * u53(&seed));
        }
        if (!same(smz_glibc_pow(x, y), pow(x, y))) { if (!wrong && bad) { bad[0] = x; bad[1] = y; } wrong++; }
    }
    return wrong;
}
