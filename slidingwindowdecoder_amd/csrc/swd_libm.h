// exp() and log1p() evaluated exactly like the C library the reference links against, so that the quaternary
// decoder's posteriors (bp4_osd.pyx:533-589 -> log1pexp / logaddexp, src/include/bpgd.cpp:399-416) come out
// bit-identical on the device instead of "equal up to the math library".
//
// The reference calls std::exp / std::log1p of glibc (third-party, not in /root/reference; the goldens were
// recorded with Ubuntu GLIBC 2.35 on an FMA-capable x86-64, where the dynamic linker selects the FMA build):
//   exp    sysdeps/ieee754/dbl-64/e_exp.c (table-driven, N = 128, degree-5 polynomial; from ARM optimized-routines).
//          Which operations the FMA build fuses was read off the disassembly of that libm:
//            kd = fma(x, N/ln2, shift); r = fma(kd, -ln2lo/N, fma(kd, -ln2hi/N, x));
//            tmp = fma(r2*r2, fma(r, C5, C4), fma(r2, fma(r, C3, C2), tail + r)); result = fma(scale, tmp, scale)
//          (the subnormal-result branch keeps scale + scale*tmp unfused).
//   log1p  sysdeps/ieee754/dbl-64/s_log1p.c (fdlibm; no FMA build in 2.35, plain IEEE operations).
// tests/test_libm_restatement.py compiles this header with gcc and holds both functions to the host libm bit for
// bit on random arguments.  Everything here is plain IEEE-754 double arithmetic: build with -ffp-contract=off.
#pragma once
#include <stdint.h>

#include "swd_exp_table.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SWD_LIBM_FN __host__ __device__ __forceinline__
#else
#define SWD_LIBM_FN static inline
#endif

#if defined(__HIPCC__)
static __device__ const uint64_t swd_exp_tab_dev[256] = {SWD_EXP_TABLE_VALUES};
#endif
static const uint64_t swd_exp_tab_host[256] = {SWD_EXP_TABLE_VALUES};

SWD_LIBM_FN uint64_t swd_asu(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
SWD_LIBM_FN double swd_asd(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }
SWD_LIBM_FN uint64_t swd_exp_tab(unsigned i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return swd_exp_tab_dev[i];
#else
    return swd_exp_tab_host[i];
#endif
}

// `tab`: where the 256-word table is read from -- NULL: the constant array above; the quaternary decoder's BP kernel passes its
// copy in LDS (one 16-byte LDS read per evaluation instead of two dependent gathers through the vector memory path)
SWD_LIBM_FN double swd_exp_from(double x, const uint64_t *tab) {
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3,
                 C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    uint32_t abstop = (uint32_t)(swd_asu(x) >> 52) & 0x7ffu;
    if (abstop - 0x3c9u >= 0x408u - 0x3c9u) {          // |x| < 2^-54 or |x| >= 512
        if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;
        if (abstop >= 0x409u) {                        // |x| >= 1024, inf, nan
            if (swd_asu(x) == 0xfff0000000000000ull) return 0.0;
            if (abstop >= 0x7ffu) return 1.0 + x;
            return (swd_asu(x) >> 63) ? 0.0 : swd_asd(0x7ff0000000000000ull); // underflow / overflow
        }
        abstop = 0;
    }
    double kd = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = swd_asu(kd);
    kd -= Shift;
    const double r = __builtin_fma(kd, NegLn2loN, __builtin_fma(kd, NegLn2hiN, x));
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const uint64_t top = ki << 45;
    const double tail = swd_asd(tab ? tab[idx] : swd_exp_tab(idx));
    uint64_t sbits = (tab ? tab[idx + 1] : swd_exp_tab(idx + 1)) + top;
    const double r2 = r * r;
    const double tmp = __builtin_fma(r2 * r2, __builtin_fma(r, C5, C4), __builtin_fma(r2, __builtin_fma(r, C3, C2), tail + r));
    if (abstop == 0) {                                 // 512 <= |x| < 1024: the scale factor needs care
        if ((ki & 0x80000000ull) == 0) {               // k > 0: result may overflow
            sbits -= 1009ull << 52;
            const double scale = swd_asd(sbits);
            return 0x1p1009 * __builtin_fma(scale, tmp, scale);
        }
        sbits += 1022ull << 52;                        // k < 0: result may be subnormal
        const double scale = swd_asd(sbits);
        const double st = scale * tmp;
        double y = scale + st;
        if (y < 1.0) {
            double lo = scale - y + st;
            const double hi = 1.0 + y;
            lo = 1.0 - hi + y + lo;
            y = (hi + lo) - 1.0;
            if (y == 0.0) y = 0.0;
        }
        return 0x1p-1022 * y;
    }
    const double scale = swd_asd(sbits);
    return __builtin_fma(scale, tmp, scale);
}

SWD_LIBM_FN double swd_exp(double x) { return swd_exp_from(x, (const uint64_t *)0); }

SWD_LIBM_FN double swd_log1p(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double f = 0.0, c = 0.0, u;
    const int32_t hx = (int32_t)(swd_asu(x) >> 32), ax = hx & 0x7fffffff;
    int32_t k = 1, hu = 0;
    if (hx < 0x3FDA827A) {                                       // x < 0.41422
        if (ax >= 0x3ff00000) {                                  // x <= -1
            if (x == -1.0) return swd_asd(0xfff0000000000000ull);
            return swd_asd(0x7ff8000000000000ull);
        }
        if (ax < 0x3e200000) return (ax < 0x3c900000) ? x : x - x * x * 0.5; // |x| < 2^-29
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) { k = 0; f = x; hu = 1; }  // -0.2929 < x < 0.41422
    } else if (hx >= 0x7ff00000) return x + x;
    if (k != 0) {
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = (int32_t)(swd_asu(u) >> 32);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);         // correction term
            c /= u;
        } else {
            u = x;
            hu = (int32_t)(swd_asu(u) >> 32);
            k = (hu >> 20) - 1023;
            c = 0.0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = swd_asd((swd_asu(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));
        } else {
            k += 1;
            u = swd_asd((swd_asu(u) & 0xffffffffull) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    const double hfsq = 0.5 * f * f;
    if (hu == 0) {                                               // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c += k * ln2_lo;
            return k * ln2_hi + c;
        }
        const double R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f), z = s * s;
    const double R1 = z * Lp1, z2 = z * z, R2 = Lp2 + z * Lp3, z4 = z2 * z2, R3 = Lp4 + z * Lp5, z6 = z4 * z2, R4 = Lp6 + z * Lp7;
    const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}
