// glibc's double-precision log() and pow(), restated operation by operation -- for the Dirichlet root noise only.
//
// Why this exists.  The reference draws its root noise with np.random.dirichlet (monte_carlo_tree_search.py:220); numpy's
// legacy gamma sampler (numpy/random/src/legacy/legacy-distributions.c: legacy_standard_gamma, legacy_standard_exponential)
// calls libm's scalar log() and pow().  Those are not correctly rounded, so "a log that is right to 1 ulp" (the device
// library's) yields root priors that differ from the reference's in the last place for ~3-7 % of the trees (rounds 1-5:
// priors held to 1e-13).  glibc >= 2.28 implements both as table-driven sequences of IEEE operations (sysdeps/ieee754/dbl-64/
// e_log.c, e_pow.c: the ARM Optimized Routines 18.11 algorithms); restating THAT sequence with the same tables makes the device's
// noise -- and the float64 root priors -- the reference's bit for bit.
//
// What exactly is restated.  On x86-64, libm resolves log / pow to its FMA builds (__log_fma / __pow_fma: the generic C code
// compiled with -mfma -mavx2, selected by an ifunc on every CPU with FMA + AVX2, i.e. every x86 server of the last decade;
// numpy's and the oracle's calls land there in this image and on the GPU box).  Which multiply-add pairs are fused there is the
// compiler's choice (__builtin_fma where the source asks + GCC's default contraction), so this file follows the MACHINE CODE of
// this image's libm.so.6 (Ubuntu GLIBC 2.35-0ubuntu3.11; `objdump -d` of the functions behind log@@GLIBC_2.29 and
// pow@@GLIBC_2.29): every fma() below is a vfmadd/vfmsub/vfnmadd there, every separate * or + a vmulsd / vaddsd / vsubsd, in
// the same association.  The translation units that include this file are compiled with -ffp-contract=off, so nothing is
// fused or split behind its back; all operations are IEEE-754 double (v_fma_f64 / v_mul_f64 / v_add_f64 on gfx950).
// Tables: smz_glibc_tables.inc, read out of the same libm.so.6 by tools/gen_glibc_tables.py.
//
// Pinned on the CPU (tests/test_glibc_math.py, `-m "not gpu"`): the same source, compiled by gcc into oracle/libglibccheck.so,
// against libm's own log / pow on > 10^7 arguments of the sampler's ranges and on the special-case boundaries; on the GPU
// (tests/test_gpu_glibc_math.py) the device build against the host's libm through smz_debug_glibc_log_pow.
//
// Domain.  smz_glibc_log(x): any double (non-positive / non-finite arguments return what libm returns, without errno).
// smz_glibc_pow(x, y): x >= +0 finite or +inf, y > 0 finite -- what the gamma sampler can pass (U^(1/shape),
// (1 - shape + shape Y)^(1/shape) with 0 < shape <= 1); anything else returns NaN so that a misuse is loud.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SMZ_GLIBC_FN __device__ static inline
#define SMZ_GLIBC_TABLE(name, n) static __device__ const uint64_t name[n]
#else
#define SMZ_GLIBC_FN static inline
#define SMZ_GLIBC_TABLE(name, n) static const uint64_t name[n]
#endif

#include "smz_glibc_tables.inc"

SMZ_GLIBC_FN double smz_gl_f64(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
SMZ_GLIBC_FN uint64_t smz_gl_u64(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
#define SMZ_GL_FMA(a, b, c) __builtin_fma((a), (b), (c))

// e_log.c: __log (the FMA build).  LOG_TABLE_BITS 7, OFF 0x3fe6000000000000.
SMZ_GLIBC_FN double smz_glibc_log(double x) {
    const uint64_t *D = smz_glibc_log_data;
    uint64_t ix = smz_gl_u64(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    // 1 - 2^-4 <= x < 1 + 0x1.09p-4: log1p polynomial on r = x - 1 with an exact hi / lo split of r - r^2 / 2
    if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double B0 = smz_gl_f64(D[7]);
        double p2 = SMZ_GL_FMA(r, smz_gl_f64(D[9]), smz_gl_f64(D[8]));      // B1 + r B2
        double p5 = SMZ_GL_FMA(r, smz_gl_f64(D[12]), smz_gl_f64(D[11]));    // B4 + r B5
        const double r2 = r * r;
        double p8 = SMZ_GL_FMA(r, smz_gl_f64(D[15]), smz_gl_f64(D[14]));    // B7 + r B8
        p2 = SMZ_GL_FMA(r2, smz_gl_f64(D[10]), p2);                         // + r2 B3
        p5 = SMZ_GL_FMA(r2, smz_gl_f64(D[13]), p5);                         // + r2 B6
        const double r3 = r * r2;
        double p = SMZ_GL_FMA(r2, smz_gl_f64(D[16]), p8);                   // B7 + r B8 + r2 B9
        p = SMZ_GL_FMA(r3, smz_gl_f64(D[17]), p);                           // + r3 B10
        p = SMZ_GL_FMA(p, r3, p5);
        p = SMZ_GL_FMA(p, r3, p2);                                          // y = r3 * p, folded into the last fma below
        const double two27 = 0x1p27;
        const double rw = SMZ_GL_FMA(r, two27, r);                          // r + r * 2^27
        const double rhi = SMZ_GL_FMA(-two27, r, rw);                       // (r + w) - w
        const double rhi2 = rhi * rhi;
        const double rlo = r - rhi;
        const double hi = SMZ_GL_FMA(rhi2, B0, r);                          // r + rhi^2 * B0
        const double rmh = r - hi;
        const double rpr = r + rhi;
        double lo = SMZ_GL_FMA(rhi2, B0, rmh);                              // r - hi + rhi^2 * B0
        const double brl = B0 * rlo;
        lo = SMZ_GL_FMA(brl, rpr, lo);                                      // + B0 rlo (rhi + r)
        const double y = SMZ_GL_FMA(p, r3, lo);
        return hi + y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        // x < 2^-1022, infinite or NaN
        if (ix * 2 == 0) return -__builtin_inf();                            // log(+-0) = -inf (divide-by-zero)
        if (ix == 0x7ff0000000000000ull) return x;                           // log(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __builtin_nan("");   // x < 0 or NaN
        ix = smz_gl_u64(x * 0x1p52);                             // subnormal: scale by 2^52
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = smz_gl_f64(D[18 + 2 * i]), logc = smz_gl_f64(D[19 + 2 * i]);
    const double z = smz_gl_f64(iz);
    const double kd = (double)k;
    const double r = SMZ_GL_FMA(z, invc, -1.0);
    const double w = SMZ_GL_FMA(kd, smz_gl_f64(D[0]), logc);                 // kd Ln2hi + logc
    const double q12 = SMZ_GL_FMA(r, smz_gl_f64(D[4]), smz_gl_f64(D[3]));    // A1 + r A2
    const double hi = r + w;
    const double r2 = r * r;
    double lo = w - hi;
    lo = lo + r;
    lo = SMZ_GL_FMA(kd, smz_gl_f64(D[1]), lo);                               // + kd Ln2lo
    const double r3 = r * r2;
    double q = SMZ_GL_FMA(r, smz_gl_f64(D[6]), smz_gl_f64(D[5]));            // A3 + r A4
    lo = SMZ_GL_FMA(r2, smz_gl_f64(D[2]), lo);                               // + r2 A0
    q = SMZ_GL_FMA(q, r2, q12);
    const double y = SMZ_GL_FMA(r3, q, lo);
    return y + hi;
}

// e_pow.c: exp_inline's special case (the scale 2^(k/N) under- or overflowed; |x| >= 512 only)
SMZ_GLIBC_FN double smz_glibc_exp_specialcase(double tmp, uint64_t sbits, uint64_t ki) {
    if ((ki & 0x80000000ull) == 0) {
        sbits -= 1009ull << 52;                                              // k > 0: the result may overflow
        const double scale = smz_gl_f64(sbits);
        const double y = SMZ_GL_FMA(scale, tmp, scale);
        return y * 0x1p1009;
    }
    sbits += 1022ull << 52;                                                  // k < 0: care in the subnormal range
    const double scale = smz_gl_f64(sbits);
    const double st = tmp * scale;
    double y = scale + st;
    if (__builtin_fabs(y) < 1.0) {
        const double one = y < 0.0 ? -1.0 : 1.0;
        double lo = scale - y;
        lo = lo + st;
        const double hi = y + one;
        double t = one - hi;
        t = t + y;
        t = t + lo;
        t = t + hi;
        y = t - one;
        if (y == 0.0) y = smz_gl_f64(sbits & 0x8000000000000000ull);
    }
    return y * 0x1p-1022;
}

// e_pow.c: __pow (the FMA build) = log_inline (hi + lo to ~68 bits) -> y * (hi + lo) -> exp_inline.
SMZ_GLIBC_FN double smz_glibc_pow(double x, double y) {
    const uint64_t *L = smz_glibc_pow_log_data;
    const uint64_t *E = smz_glibc_exp_data;
    uint64_t ix = smz_gl_u64(x);
    const uint64_t iy = smz_gl_u64(y);
    const uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
    if ((ix >> 63) || (iy >> 63) || iy * 2 == 0 || topy >= 0x7ffu || (topx == 0x7ffu && ix != 0x7ff0000000000000ull))
        return __builtin_nan("");                                            // outside the sampler's domain (see the header)
    if (topx - 1u >= 0x7ffu - 1u || topy - 0x3beu >= 0x80u) {
        if (ix * 2 == 0 || ix == 0x7ff0000000000000ull) return x * x;        // (+0)^y = 0, inf^y = inf for y > 0
        if (topy - 0x3beu >= 0x80u) {
            if (ix == 0x3ff0000000000000ull) return 1.0;
            if (topy < 0x3beu) return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;     // |y| < 2^-65: 1 +- tiny
            return ix > 0x3ff0000000000000ull ? __builtin_inf() : 0.0;        // y >= 2^63: overflow / underflow
        }
        if (topx == 0) {                                                     // subnormal x: normalise
            ix = smz_gl_u64(x * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    // ---- log_inline: POW_LOG_TABLE_BITS 7, OFF 0x3fe6955500000000
    const uint64_t tmp = ix - 0x3fe6955500000000ull;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double z = smz_gl_f64(iz);
    const double kd = (double)k;
    const double invc = smz_gl_f64(L[9 + 4 * i]), logc = smz_gl_f64(L[11 + 4 * i]), logctail = smz_gl_f64(L[12 + 4 * i]);
    const double t1 = SMZ_GL_FMA(kd, smz_gl_f64(L[0]), logc);                 // kd Ln2hi + logc
    const double r = SMZ_GL_FMA(z, invc, -1.0);
    const double ar = r * smz_gl_f64(L[2]);                                   // A0 r   (A0 = -0.5)
    const double lo1 = SMZ_GL_FMA(kd, smz_gl_f64(L[1]), logctail);            // kd Ln2lo + logctail
    const double a12 = SMZ_GL_FMA(r, smz_gl_f64(L[4]), smz_gl_f64(L[3]));     // A1 + r A2
    const double a34 = SMZ_GL_FMA(r, smz_gl_f64(L[6]), smz_gl_f64(L[5]));     // A3 + r A4
    const double t2 = r + t1;
    const double ar2 = r * ar;
    const double t1m = t1 - t2;
    const double ar3 = r * ar2;
    const double lo3 = SMZ_GL_FMA(ar, r, -ar2);
    const double lo2 = t1m + r;
    double a = SMZ_GL_FMA(r, smz_gl_f64(L[8]), smz_gl_f64(L[7]));            // A5 + r A6
    const double lhi = t2 + ar2;
    double lo4 = t2 - lhi;
    a = SMZ_GL_FMA(a, ar2, a34);
    lo4 = lo4 + ar2;
    a = SMZ_GL_FMA(ar2, a, a12);
    double lo = lo1 + lo2;
    lo = lo + lo3;
    lo = lo + lo4;
    lo = SMZ_GL_FMA(ar3, a, lo);                                             // + p
    const double hi = lhi + lo;
    double tail = lhi - hi;
    tail = tail + lo;
    // ---- y * (hi + tail)
    const double ehi = y * hi;
    const double em = SMZ_GL_FMA(hi, y, -ehi);
    const double elo = SMZ_GL_FMA(y, tail, em);
    // ---- exp_inline(ehi, elo, sign_bias = 0): EXP_TABLE_BITS 7
    uint32_t abstop = (uint32_t)(smz_gl_u64(ehi) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x3fu) {
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + ehi;                // |ehi| < 2^-54
        if (abstop >= 0x409u) return (smz_gl_u64(ehi) >> 63) ? 0.0 : __builtin_inf();   // |ehi| >= 1024: under- / overflow
        abstop = 0;                                                          // 512 <= |ehi| < 1024: special-cased below
    }
    const double shift = smz_gl_f64(E[1]);
    double kd2 = SMZ_GL_FMA(ehi, smz_gl_f64(E[0]), shift);                   // ehi InvLn2N + Shift
    const uint64_t ki = smz_gl_u64(kd2);
    kd2 = kd2 - shift;
    double rr = SMZ_GL_FMA(kd2, smz_gl_f64(E[2]), ehi);                      // ehi + kd NegLn2hiN
    rr = SMZ_GL_FMA(kd2, smz_gl_f64(E[3]), rr);                              // + kd NegLn2loN
    const int idx = (int)(ki & 127);
    const uint64_t sbits = E[15 + 2 * idx] + (ki << 45);
    rr = elo + rr;
    const double c23 = SMZ_GL_FMA(rr, smz_gl_f64(E[5]), smz_gl_f64(E[4]));   // C2 + r C3
    const double tr = rr + smz_gl_f64(E[14 + 2 * idx]);                      // tail + r
    const double rr2 = rr * rr;
    const double c45 = SMZ_GL_FMA(rr, smz_gl_f64(E[7]), smz_gl_f64(E[6]));   // C4 + r C5
    double t = SMZ_GL_FMA(c23, rr2, tr);
    const double rr4 = rr2 * rr2;
    t = SMZ_GL_FMA(c45, rr4, t);
    if (abstop == 0) return smz_glibc_exp_specialcase(t, sbits, ki);
    const double scale = smz_gl_f64(sbits);
    return SMZ_GL_FMA(t, scale, scale);
}
