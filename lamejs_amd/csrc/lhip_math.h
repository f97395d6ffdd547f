// Device math that must agree bit-for-bit with the reference's host engine (V8 7.8 `Math.*`,
// v8/src/base/ieee754.cc, fdlibm lineage -- third party, not part of /root/reference):
//   Math.log10(x)       -> v8_log10()   used by mask_add, calc_noise, athAdjust
//   Math.pow(10, y)     -> v8_pow10()   used by athAdjust
//   Math.pow(x, 0.5)    -> sqrt(x)      (fdlibm/V8 shortcut for y == 0.5)
//   (0 | x), Int32Array -> js_toint32()
// Only f64 +,-,*,/ and integer bit manipulation are used; no FMA (-ffp-contract=off).
// v8_pow10 exploits that the base is the constant 10: the first half of the fdlibm pow
// algorithm (log2(x) as a hi/lo pair) depends on x alone, so it is evaluated once on the
// host (pow_log2_parts) and the device only runs the y-dependent half.
#pragma once
#include "lhip_defs.h"

namespace lhip {

#ifdef LHIP_HOSTSIM
LHIP_DEV uint32_t d_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
LHIP_DEV uint32_t d_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
LHIP_DEV double d_make(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
LHIP_DEV double d_sqrt(double x) { return sqrt(x); }
LHIP_DEV double d_abs(double x) { return fabs(x); }
#else
LHIP_DEV uint32_t d_hi(double x) { return (uint32_t)__double2hiint(x); }
LHIP_DEV uint32_t d_lo(double x) { return (uint32_t)__double2loint(x); }
LHIP_DEV double d_make(uint32_t hi, uint32_t lo) { return __hiloint2double((int)hi, (int)lo); }
LHIP_DEV double d_sqrt(double x) { return __builtin_sqrt(x); }   // correctly rounded f64 sqrt
LHIP_DEV double d_abs(double x) { return __builtin_fabs(x); }
#endif
LHIP_DEV double d_with_hi(double x, uint32_t hi) { return d_make(hi, d_lo(x)); }
LHIP_DEV double d_trunc_lo(double x) { return d_make(d_hi(x), 0u); }

// ECMAScript ToInt32 for the values this path produces (finite, |x| < 2^31 in practice).
LHIP_DEV int32_t js_toint32(double d) {
    if (d >= -2147483648.0 && d <= 2147483647.0) return (int32_t)d;        // the only case frame material produces (NaN compares false)
    // ECMAScript ToInt32 for every other double: the integer part modulo 2^32, from the bits (NaN and the infinities give 0)
    const uint32_t hi = d_hi(d), lo = d_lo(d);
    const int32_t e = (int32_t)((hi >> 20) & 0x7ffu);
    if (e == 0x7ff) return 0;
    const uint64_t mant = (((uint64_t)(hi & 0x000fffffu) | 0x00100000ull) << 32) | lo;     // |d| = mant * 2^(e - 1075), e >= 1054 here
    const int32_t sh = e - 1075;
    uint32_t m = 0;
    if (sh >= 32) m = 0;
    else if (sh >= 0) m = (uint32_t)(mant << sh);
    else m = (uint32_t)(mant >> (-sh));
    return (int32_t)((hi >> 31) ? (0u - m) : m);
}

// natural logarithm, fdlibm method: x = 2^k (1+f); log(1+f) = f - s*(f - R(s^2)) ..., Remez poly
LHIP_DEV double v8_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 two54 = 1.80143985094819840000e+16;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    int32_t hx = (int32_t)d_hi(x);
    uint32_t lx = d_lo(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        hx = (int32_t)d_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    int32_t i = (hx + 0x95f64) & 0x100000;
    x = d_with_hi(x, (uint32_t)(hx | (i ^ 0x3ff00000)));
    k += (i >> 20);
    double f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            double dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        double R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        double dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    double dk = (double)k;
    double z = s * s;
    i = hx - 0x6147a;
    double w = z * z;
    int32_t j = 0x6b851 - hx;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    double R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

LHIP_DEV double v8_log10(double x) {
    const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
                 log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    int32_t hx = (int32_t)d_hi(x);
    uint32_t lx = d_lo(x);
    int32_t k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / 0.0;
        if (hx < 0) return (x - x) / 0.0;
        k -= 54;
        x *= two54;
        hx = (int32_t)d_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    int32_t i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    double y = (double)(k + i);
    x = d_with_hi(x, (uint32_t)hx);
    double z = y * log10_2lo + ivln10 * v8_log(x);
    return z + y * log10_2hi;
}

// ---- the two truncations of quantize_lines_xrpow (Takehiro.js:125-165) -----------------------------------------------------
//     rx = (int)(x * istep)                 x, istep: Float32Array values, the product is formed in f64 (exact: 24 x 24 bits)
//     ix = (int)(x * istep + adj43[rx])     adj43: Float32Array value in (0, 0.5); one f64 rounding
// with 0 <= x * istep <= 8206 (count_bits' guard).  On the device they are ONE f32 instruction each under round-toward-zero
// (v_mul_f32 / v_fma_f32, then the truncating v_cvt_i32_f32) instead of f64 conversions, multiply and add (quarter-rate ops):
//   * floor(trunc24(p)) == floor(p) for an exact value 0 <= p < 2^24, because floor(p) is itself a 24-bit number <= p;
//   * the exact sum S = x * istep + adj is a multiple of g = min(ulp(x) ulp(istep), ulp(adj)) >= 2^-25 resp. 2^(E - 46) (E the
//     exponent sum), while the f64 rounding error of S is at most 2^(E + 2 - 53) and at most 2^-40: smaller than g, so RN64(S)
//     cannot reach the next integer above S unless S is that integer; hence floor(RN64(S)) == floor(S) == floor(trunc24(S)).
// FP_ROUND's single-precision field is switched inside the asm statement, so no other instruction can see the changed mode (the
// f64 field is untouched).  Host builds evaluate the reference's f64 expressions literally; tests: device math op 8.
template <int N> LHIP_DEV void q_floor_prod(const float (&xa)[N], const float (&xb)[N], float istep, int (&ra)[N], int (&rb)[N]) {
#ifdef LHIP_HOSTSIM
    for (int j = 0; j < N; j++) { ra[j] = (int)((double)xa[j] * (double)istep); rb[j] = (int)((double)xb[j] * (double)istep); }
#else
    static_assert(N == 5, "device form is written for 5 pairs per lane");
    float a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
                 "v_mul_f32 %0, %10, %20\n\tv_mul_f32 %1, %11, %20\n\tv_mul_f32 %2, %12, %20\n\tv_mul_f32 %3, %13, %20\n\tv_mul_f32 %4, %14, %20\n\t"
                 "v_mul_f32 %5, %15, %20\n\tv_mul_f32 %6, %16, %20\n\tv_mul_f32 %7, %17, %20\n\tv_mul_f32 %8, %18, %20\n\tv_mul_f32 %9, %19, %20\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4)
                 : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xa[4]), "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]), "v"(xb[4]), "v"(istep));
    ra[0] = (int)a0; ra[1] = (int)a1; ra[2] = (int)a2; ra[3] = (int)a3; ra[4] = (int)a4;
    rb[0] = (int)b0; rb[1] = (int)b1; rb[2] = (int)b2; rb[3] = (int)b3; rb[4] = (int)b4;
#endif
}
template <int N> LHIP_DEV void q_floor_fma(const float (&xa)[N], const float (&xb)[N], float istep, const float (&ja)[N], const float (&jb)[N],
                                           int (&va)[N], int (&vb)[N]) {
#ifdef LHIP_HOSTSIM
    for (int j = 0; j < N; j++) {
        va[j] = (int)((double)xa[j] * (double)istep + (double)ja[j]);
        vb[j] = (int)((double)xb[j] * (double)istep + (double)jb[j]);
    }
#else
    static_assert(N == 5, "device form is written for 5 pairs per lane");
    float a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\t"
                 "v_fma_f32 %0, %10, %20, %21\n\tv_fma_f32 %1, %11, %20, %22\n\tv_fma_f32 %2, %12, %20, %23\n\tv_fma_f32 %3, %13, %20, %24\n\tv_fma_f32 %4, %14, %20, %25\n\t"
                 "v_fma_f32 %5, %15, %20, %26\n\tv_fma_f32 %6, %16, %20, %27\n\tv_fma_f32 %7, %17, %20, %28\n\tv_fma_f32 %8, %18, %20, %29\n\tv_fma_f32 %9, %19, %20, %30\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4)
                 : "v"(xa[0]), "v"(xa[1]), "v"(xa[2]), "v"(xa[3]), "v"(xa[4]), "v"(xb[0]), "v"(xb[1]), "v"(xb[2]), "v"(xb[3]), "v"(xb[4]), "v"(istep),
                   "v"(ja[0]), "v"(ja[1]), "v"(ja[2]), "v"(ja[3]), "v"(ja[4]), "v"(jb[0]), "v"(jb[1]), "v"(jb[2]), "v"(jb[3]), "v"(jb[4]));
    va[0] = (int)a0; va[1] = (int)a1; va[2] = (int)a2; va[3] = (int)a3; va[4] = (int)a4;
    vb[0] = (int)b0; vb[1] = (int)b1; vb[2] = (int)b2; vb[3] = (int)b3; vb[4] = (int)b4;
#endif
}

// v8_log10 for positive normal operands, +inf and NaN (x >= 2^-1022 or NaN) without data-dependent branches: in a wave program
// every two-sided `if` on a per-lane value costs exec-mask bookkeeping on the (per-CU) scalar unit and both sides run anyway.
// Same operations in the same order as v8_log10 / v8_log above -- the four return expressions of v8_log's main path and the two
// of its short series (|f| < 2^-20 after range reduction; the f == 0 returns are the short-series expressions with R = 0, bit
// for bit) -- selected per lane.  Zero, negative and subnormal operands are NOT handled (callers clamp: calc_noise passes
// max(noise, 1e-20)).  Bit-identical to v8_log10 on its domain (tests: device math op 7, 60 M operands on the host).
#if defined(LHIP_EXP_CONSTBLOCK) && !defined(LHIP_HOSTSIM)
// A/B build (round 6): the routine's twelve f64 constants as one block in constant memory -- a scalar load of the block instead of two s_mov_b32 literals per constant
__constant__ double LHIP_LOG10_K[12] = {4.34294481903251816668e-01, 3.01029995663611771306e-01, 3.69423907715893078616e-13, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};
#endif
LHIP_DEV double v8_log10_pos(double x) {
#if defined(LHIP_EXP_CONSTBLOCK) && !defined(LHIP_HOSTSIM)
    const double* K_ = LHIP_LOG10_K;
    const double ivln10 = K_[0], log10_2hi = K_[1], log10_2lo = K_[2], ln2_hi = K_[3], ln2_lo = K_[4], Lg1 = K_[5], Lg2 = K_[6], Lg3 = K_[7], Lg4 = K_[8], Lg5 = K_[9], Lg6 = K_[10], Lg7 = K_[11];
#else
    const double ivln10 = 4.34294481903251816668e-01, log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
#endif
    int32_t hx = (int32_t)d_hi(x);
    const bool infnan = hx >= 0x7ff00000;
    // log10: x = 2^k10 * m, operand of log in [0.5, 2)
    const int32_t k10 = (hx >> 20) - 1023;
    const int32_t i10 = (int32_t)(((uint32_t)k10 & 0x80000000u) >> 31);
    const double y = (double)(k10 + i10);
    hx = (hx & 0x000fffff) | ((0x3ff - i10) << 20);
    // log of that operand
    int32_t k = (hx >> 20) - 1023;                          // -1 or 0
    hx &= 0x000fffff;
    const int32_t i = (hx + 0x95f64) & 0x100000;
    const double xm = d_make((uint32_t)(hx | (i ^ 0x3ff00000)), d_lo(x));
    k += (i >> 20);                                         // -1, 0, 1
    const double f = xm - 1.0;
    const bool small = (0x000fffff & (2 + hx)) < 3;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const int32_t i2 = (hx - 0x6147a) | (0x6b851 - hx);
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double ff = f * f;
    const double hfsq = 0.5 * ff;
    const double Rs = ff * (0.5 - 0.33333333333333333 * f);
    const bool big = i2 > 0, k0 = k == 0;
    const double v = s * (big ? hfsq + R : f - R);
    const double dl = dk * ln2_lo, dh = dk * ln2_hi;
    // k == 0:  f - R' with R' = Rs | hfsq - v | v ;  k != 0:  dh - ((R'' ) - f) with R'' = Rs - dl | hfsq - (v + dl) | v - dl
    const double r0 = small ? Rs : (big ? hfsq - v : v);
    const double r1 = small ? Rs - dl : (big ? hfsq - (v + dl) : v - dl);
    const double lg = k0 ? f - r0 : dh - (r1 - f);
    const double zz = y * log10_2lo + ivln10 * lg;
    const double res = zz + y * log10_2hi;
    return infnan ? x + x : res;
}

// ---- calc_noise without the logarithm (QuantizePVT.js:846-868) ------------------------------------------------------------------
// Per band the reference computes noise = log10(max(x, 1e-20)), x = band noise / allowed noise, and then only asks:
//   over:  noise > 0                      <=>  x > 1   (log10 is exact in sign around 1, and so is its Float32 copy)
//   tmp:   max(ToInt32(noise * 10 + .5), 1)   -- the band's contribution tmp^2 to over_SSD
// (max_noise needs the value itself; it is only read for results without distorted bands, see q_calc_noise_).  tmp is a step
// function of x with steps at 10^((k - .5) / 10): away from the steps ANY approximation of 10 log10(x) + .5 with an error below the
// distance to the next integer gives the reference's integer -- whether the reference derives it from the f64 logarithm (a band
// evaluated in this call) or from the Float32 copy it cached (QuantizePVT.js:838-842; the copy moves noise * 10 by at most 2.4e-5).
// noise_class(x): t = log2_f32((float)x) * 10 log10(2) + .5 (error < 5e-5 for x < 1e37: one ulp of the hardware's v_log_f32 at
// |log2| <= 123, the conversion of x, the f32 multiply-add); returns tmp (0: not over), or -1 when t is within 2e-4 of an integer or x
// is outside (1, 1e37) x-range where the shortcut is valid -- the caller then evaluates the logarithm itself, as before.
#ifdef LHIP_HOSTSIM
LHIP_DEV float fast_log2f(float x) { return log2f(x); }
#else
LHIP_DEV float fast_log2f(float x) { return __builtin_amdgcn_logf(x); }       // v_log_f32
#endif
LHIP_DEV int noise_class(double x) {
    if (!(x > 1.0)) return 0;                              // not over (NaN included: the reference clamps it to 1e-20)
    if (!(x < 1e37)) return -1;
    const float t = fast_log2f((float)x) * 3.01029995663981195f + 0.5f;
    const float k = __builtin_floorf(t);
    const float fr = t - k;
    if (fr < 2e-4f || fr > 1.0f - 2e-4f) return -1;
    const int tmp = (int)k;
    return tmp < 1 ? 1 : tmp;
}
// ---- a / b where b is a Float32 value and rb = RN(1 / b) is at hand (calc_noise divides every band's noise by the same xmin in every
// call of a granule).  q0 = RN(a rb) lies within 1.5 ulp of a / b, so the residual r = a - b q0 is exact in one fma (it is a multiple of
// lsb(b) lsb(q0) below 2^26 of them), and q0 + r rb = a / b + (a / b - q0) eps with |eps| <= 2^-53: less than 2^-104 |a / b| away from
// the quotient.  A quotient by a divisor of 24 significant bits keeps at least 2^-78 |a / b| from every midpoint of the f64 grid
// (|a - m b| is a non-zero multiple of lsb(m) lsb(b); a = m b would need 54 bits), so the fma's single rounding is RN(a / b): the
// result is the division's, bit for bit.  Proven for finite a >= 0 and b a positive Float32 (subnormal ones included: normal doubles)
// whose reciprocal is finite -- recip_for_div returns 0 for any other b and the caller then divides.
// maximum of two Float32 values that are >= +0 and not NaN: IEEE order == integer order of the bit patterns (one v_max_i32; the
// compare-and-select the source form `if (v > m) m = v` compiles to costs three issue slots)
LHIP_DEV float fmax_nonneg(float a, float b) {
    int32_t ia, ib; __builtin_memcpy(&ia, &a, 4); __builtin_memcpy(&ib, &b, 4);
    const int32_t im = ia > ib ? ia : ib;
    float r; __builtin_memcpy(&r, &im, 4); return r;
}
// the next Float32 above a positive finite one
LHIP_DEV float f32_next_up(float x) { uint32_t u; __builtin_memcpy(&u, &x, 4); u += 1; float r; __builtin_memcpy(&r, &u, 4); return r; }
// 0.0 if bit k of m is set, else 1.0 -- as two integer operations on the high word (a compare-and-select pair per word otherwise)
LHIP_DEV double one_unless_bit(uint32_t m, int k) {
#ifdef LHIP_HOSTSIM
    return ((m >> k) & 1u) ? 0.0 : 1.0;
#else
    const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe((int)~m, (unsigned)k, 1u) & 0x3ff00000u;
    return __builtin_bit_cast(double, (uint64_t)hi << 32);
#endif
}
LHIP_DEV double recip_for_div(double b) { return (b > 0.0 && b < 1e300) ? 1.0 / b : 0.0; }
LHIP_DEV double div_by_f32(double a, double b, double rb) {
    const double q0 = a * rb;
    const double r = __builtin_fma(-b, q0, a);
    return __builtin_fma(r, rb, q0);
}

// the same from the logarithm (what the reference computes): nl = log10(max(x, 1e-20)) or its Float32 copy
LHIP_DEV int noise_class_of_log(double nl) {
    if (!(nl > 0.0)) return 0;
    int tmp = (int)(nl * 10 + .5);                         // 0 < nl < ~400: truncation == ToInt32
    return tmp < 1 ? 1 : tmp;
}

// ---- pow(x, y), x > 0 finite normal (x == 10 on this path), |y| < 2^31 ----
struct PowBase { double t1, t2, x; };   // log2(x) = t1 + t2, t1 with a zeroed low word; the base itself (x > 1, finite) for the special cases of y

// x-dependent half (on the host once per table set for base 10; on the device for v8_pow below).  Valid for positive normal x, any y with |y| <= 2^31.
#ifdef LHIP_HOSTSIM
static inline
#else
static __host__ __device__ __forceinline__
#endif
PowBase pow_log2_parts(double x) {
    const double bp[2] = {1.0, 1.5}, dp_h[2] = {0.0, 5.84962487220764160156e-01},
                 dp_l[2] = {0.0, 1.35003920212974897128e-08};
    const double L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
                 L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
                 cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01, cp_l = -7.02846165095275826516e-09;
    uint64_t ub; __builtin_memcpy(&ub, &x, 8);
    int32_t ix = (int32_t)(ub >> 32) & 0x7fffffff;
    int32_t n = (ix >> 20) - 0x3ff, k;
    int32_t j = ix & 0x000fffff;
    ix = j | 0x3ff00000;
    if (j <= 0x3988E) k = 0;
    else if (j < 0xBB67A) k = 1;
    else { k = 0; n += 1; ix -= 0x00100000; }
    auto mk = [](uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double d; __builtin_memcpy(&d, &u, 8); return d; };
    auto lo_of = [](double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return (uint32_t)u; };
    auto hi_of = [](double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return (uint32_t)(u >> 32); };
    double ax = mk((uint32_t)ix, (uint32_t)ub);
    double u = ax - bp[k];
    double v = 1.0 / (ax + bp[k]);
    double ss = u * v;
    double s_h = mk(hi_of(ss), 0);
    double t_h = mk((uint32_t)(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)), 0);
    double t_l = ax - (t_h - bp[k]);
    double s_l = v * ((u - s_h * t_h) - s_h * t_l);
    double s2 = ss * ss;
    double r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
    r += s_l * (s_h + ss);
    s2 = s_h * s_h;
    t_h = 3.0 + s2 + r;
    t_h = mk(hi_of(t_h), 0);
    t_l = r - ((t_h - 3.0) - s2);
    u = s_h * t_h;
    v = s_l * t_h + t_l * ss;
    double p_h = u + v;
    p_h = mk(hi_of(p_h), 0);
    double p_l = v - (p_h - u);
    double z_h = cp_h * p_h;
    double z_l = cp_l * p_h + p_l * cp + dp_l[k];
    double t = (double)n;
    PowBase pb;
    pb.t1 = (((z_h + z_l) + dp_h[k]) + t);
    pb.t1 = mk(hi_of(pb.t1), 0);
    pb.t2 = z_l - (((pb.t1 - t) - dp_h[k]) - z_h);
    pb.x = x;
    (void)lo_of;
    return pb;
}

LHIP_DEV double v8_pow_from_parts(double y, double t1, double t2);
// Math.pow(x, y) for the base the parts were made from (finite, x > 1): the special cases of y that the engine answers before its
// general algorithm (ieee754::pow: y = +-0, NaN, +-inf, +-1, 2, 0.5, |y| > 2^31), then the general algorithm.
LHIP_DEV double v8_pow_base(const PowBase& pb, double y) {
    const int32_t hy = (int32_t)d_hi(y);
    const uint32_t ly = d_lo(y);
    const int32_t iy = hy & 0x7fffffff;
    if ((iy | (int32_t)ly) == 0) return 1.0;
    if (iy > 0x7ff00000 || (iy == 0x7ff00000 && ly != 0)) return pb.x + y;                 // NaN
    if (ly == 0) {
        if (iy == 0x7ff00000) return hy >= 0 ? y : 0.0;                                    // |x| > 1: +inf -> inf, -inf -> 0
        if (iy == 0x3ff00000) return hy < 0 ? 1.0 / pb.x : pb.x;
        if (hy == 0x40000000) return pb.x * pb.x;
        if (hy == 0x3fe00000) return d_sqrt(pb.x);
    }
    if (iy > 0x41e00000) return hy > 0 ? 1.0e300 * 1.0e300 : 1.0e-300 * 1.0e-300;          // |y| > 2^31 with x > 1: overflow / underflow
    return v8_pow_from_parts(y, pb.t1, pb.t2);
}

// Math.pow(x, y) for x >= 0 finite and 0 < y < 1 (NS_INTERP of the psychoacoustic model, PsyModel.js:828-842: x a ratio of thresholds,
// y a share of the bit reservoir): the engine's shortcuts that can occur there (x = 0, x = 1, y = 0.5 -> sqrt), else its general algorithm.
LHIP_DEV double v8_pow_from_parts(double y, double t1, double t2);
LHIP_DEV double v8_pow(double x, double y) {
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    if (y == 0.5) return d_sqrt(x);
    const PowBase pb = pow_log2_parts(x);
    return v8_pow_from_parts(y, pb.t1, pb.t2);
}

// y-dependent half: 2^(y * (t1 + t2)) with the engine's exact operation order (including its
// divisor grouping in the final rational step, which differs from Sun's fdlibm).
LHIP_DEV double v8_pow_from_parts(double y, double t1, double t2) {
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
                 lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
                 ovt = 8.0085662595372944372e-0017, huge = 1.0e300, tiny = 1.0e-300;
    if (y == 0.0) return 1.0;
    double y1 = d_trunc_lo(y);
    double p_l = (y - y1) * t1 + y * t2;
    double p_h = y1 * t1;
    double z = p_l + p_h;
    int32_t j = (int32_t)d_hi(z), i = (int32_t)d_lo(z);
    if (j >= 0x40900000) {
        if (((j - 0x40900000) | i) != 0) return huge * huge;
        if (p_l + ovt > z - p_h) return huge * huge;
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {
        if (((j - (int32_t)0xc090cc00) | i) != 0) return tiny * tiny;
        if (p_l <= z - p_h) return tiny * tiny;
    }
    i = j & 0x7fffffff;
    int32_t k = (i >> 20) - 0x3ff;
    int32_t n = 0;
    if (i > 0x3fe00000) {
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        double t = d_make((uint32_t)(n & ~(0x000fffff >> k)), 0);
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    double t = d_trunc_lo(p_l + p_h);
    double u = t * lg2_h;
    double v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    double w = v - (z - u);
    t = z * z;
    double tt1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    double r = (z * tt1) / ((tt1 - 2.0) - (w + z * w));
    z = 1.0 - (r - z);
    j = (int32_t)d_hi(z);
    j += (int32_t)((uint32_t)n << 20);          // n may be negative: shift the bit pattern, not the signed value
    if ((j >> 20) <= 0) {
        // subnormal result: scale by 2^n in two exact steps (scalbn)
        const double twom54 = 5.55111512312578270212e-17;
        int32_t hz = (int32_t)d_hi(z);
        int32_t kk = ((hz & 0x7ff00000) >> 20) + n;
        if (kk <= -54) return tiny * tiny;
        kk += 54;
        z = d_with_hi(z, (uint32_t)((hz & 0x800fffff) | (int32_t)((uint32_t)kk << 20)));
        return z * twom54;
    }
    return d_with_hi(z, (uint32_t)j);
}

}  // namespace lhip
