/*
 * TEST INFRASTRUCTURE -- part of the CPU oracle, never linked into the product.
 *
 * Math.log10 / Math.pow / Math.sqrt as the reference's host engine (Node v12.22.9,
 * V8 7.8.279.23, v8/src/base/ieee754.cc -- third-party, not vendored under
 * /root/reference) evaluates them.  V8 uses the classic fdlibm algorithms
 * (e_log.c, e_log10.c, e_pow.c, Sun Microsystems 1993); they are restated here
 * from the published algorithm.  Pinned by oracle/check_v8math.c, which compares
 * millions of samples bit-for-bit with the routines exported by libnode.so.72
 * (v8::base::ieee754::{log,log10,pow}) on this image.
 *
 * Compile WITHOUT fp contraction (-ffp-contract=off): no FMA may be formed.
 */
#ifndef LO_V8MATH_H
#define LO_V8MATH_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint32_t lo_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static inline uint32_t lo_lo(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double lo_mk(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }
static inline double lo_sethi(double x, uint32_t hi) { return lo_mk(hi, lo_lo(x)); }
static inline double lo_setlo(double x, uint32_t lo) { return lo_mk(lo_hi(x), lo); }

/* natural log, fdlibm e_log.c */
static double v8_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
        two54 = 1.80143985094819840000e+16,
        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
        Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
        Lg7 = 1.479819860511658591e-01;
    const double zero = 0.0;
    double hfsq, f, s, z, R, w, t1, t2, dk;
    int32_t k, hx, i, j;
    uint32_t lx;
    hx = (int32_t)lo_hi(x); lx = lo_lo(x);
    k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero;
        if (hx < 0) return (x - x) / zero;
        k -= 54; x *= two54;
        hx = (int32_t)lo_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    x = lo_sethi(x, (uint32_t)(hx | (i ^ 0x3ff00000)));
    k += (i >> 20);
    f = x - 1.0;
    if ((0x000fffff & (2 + hx)) < 3) {
        if (f == zero) {
            if (k == 0) return zero;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    s = f / (2.0 + f);
    dk = (double)k;
    z = s * s;
    i = hx - 0x6147a;
    w = z * z;
    j = 0x6b851 - hx;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    } else {
        if (k == 0) return f - s * (f - R);
        return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
    }
}

/* fdlibm e_log10.c */
static double v8_log10(double x) {
    static const double two54 = 1.80143985094819840000e+16, ivln10 = 4.34294481903251816668e-01,
        log10_2hi = 3.01029995663611771306e-01, log10_2lo = 3.69423907715893078616e-13;
    const double zero = 0.0;
    double y, z;
    int32_t i, k, hx;
    uint32_t lx;
    hx = (int32_t)lo_hi(x); lx = lo_lo(x);
    k = 0;
    if (hx < 0x00100000) {
        if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero;
        if (hx < 0) return (x - x) / zero;
        k -= 54; x *= two54;
        hx = (int32_t)lo_hi(x);
    }
    if (hx >= 0x7ff00000) return x + x;
    k += (hx >> 20) - 1023;
    i = (int32_t)(((uint32_t)k & 0x80000000u) >> 31);
    hx = (hx & 0x000fffff) | ((0x3ff - i) << 20);
    y = (double)(k + i);
    x = lo_sethi(x, (uint32_t)hx);
    z = y * log10_2lo + ivln10 * v8_log(x);
    return z + y * log10_2hi;
}

/* fdlibm s_scalbn.c (used by pow) */
static double v8_scalbn(double x, int n) {
    static const double two54 = 1.80143985094819840000e+16, twom54 = 5.55111512312578270212e-17,
        huge = 1.0e+300, tiny = 1.0e-300;
    int32_t k, hx, lx;
    hx = (int32_t)lo_hi(x); lx = (int32_t)lo_lo(x);
    k = (hx & 0x7ff00000) >> 20;
    if (k == 0) {
        if ((lx | (hx & 0x7fffffff)) == 0) return x;
        x *= two54;
        hx = (int32_t)lo_hi(x);
        k = ((hx & 0x7ff00000) >> 20) - 54;
        if (n < -50000) return tiny * x;
    }
    if (k == 0x7ff) return x + x;
    k = k + n;
    if (k > 0x7fe) return huge * copysign(huge, x);
    if (k > 0) { return lo_sethi(x, (uint32_t)((hx & 0x800fffff) | (k << 20))); }
    if (k <= -54) {
        if (n > 50000) return huge * copysign(huge, x);
        return tiny * copysign(tiny, x);
    }
    k += 54;
    x = lo_sethi(x, (uint32_t)((hx & 0x800fffff) | (k << 20)));
    return x * twom54;
}

/* fdlibm e_pow.c */
static double v8_pow(double x, double y) {
    static const double bp[] = {1.0, 1.5}, dp_h[] = {0.0, 5.84962487220764160156e-01},
        dp_l[] = {0.0, 1.35003920212974897128e-08},
        zero = 0.0, one = 1.0, two = 2.0, two53 = 9007199254740992.0, huge = 1.0e300, tiny = 1.0e-300,
        L1 = 5.99999999999994648725e-01, L2 = 4.28571428578550184252e-01, L3 = 3.33333329818377432918e-01,
        L4 = 2.72728123808534006489e-01, L5 = 2.30660745775561754067e-01, L6 = 2.06975017800338417784e-01,
        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08,
        lg2 = 6.93147180559945286227e-01, lg2_h = 6.93147182464599609375e-01, lg2_l = -1.90465429995776804525e-09,
        ovt = 8.0085662595372944372e-0017, cp = 9.61796693925975554329e-01, cp_h = 9.61796700954437255859e-01,
        cp_l = -7.02846165095275826516e-09, ivln2 = 1.44269504088896338700e+00,
        ivln2_h = 1.44269502162933349609e+00, ivln2_l = 1.92596299112661746887e-08;
    double z, ax, z_h, z_l, p_h, p_l;
    double y1, t1, t2, r, s, t, u, v, w;
    int32_t i, j, k, yisint, n;
    int32_t hx, hy, ix, iy;
    uint32_t lx, ly;

    hx = (int32_t)lo_hi(x); lx = lo_lo(x);
    hy = (int32_t)lo_hi(y); ly = lo_lo(y);
    ix = hx & 0x7fffffff; iy = hy & 0x7fffffff;

    if ((iy | ly) == 0) return one;
    if (ix > 0x7ff00000 || ((ix == 0x7ff00000) && (lx != 0)) || iy > 0x7ff00000 || ((iy == 0x7ff00000) && (ly != 0)))
        return x + y;

    yisint = 0;
    if (hx < 0) {
        if (iy >= 0x43400000) yisint = 2;
        else if (iy >= 0x3ff00000) {
            k = (iy >> 20) - 0x3ff;
            if (k > 20) { j = (int32_t)(ly >> (52 - k)); if (((uint32_t)j << (52 - k)) == ly) yisint = 2 - (j & 1); }
            else if (ly == 0) { j = iy >> (20 - k); if ((j << (20 - k)) == iy) yisint = 2 - (j & 1); }
        }
    }
    if (ly == 0) {
        if (iy == 0x7ff00000) {
            if (((ix - 0x3ff00000) | lx) == 0) return y - y;
            else if (ix >= 0x3ff00000) return (hy >= 0) ? y : zero;
            else return (hy < 0) ? -y : zero;
        }
        if (iy == 0x3ff00000) { if (hy < 0) return one / x; else return x; }
        if (hy == 0x40000000) return x * x;
        if (hy == 0x3fe00000) { if (hx >= 0) return sqrt(x); }
    }
    ax = fabs(x);
    if (lx == 0) {
        if (ix == 0x7ff00000 || ix == 0 || ix == 0x3ff00000) {
            z = ax;
            if (hy < 0) z = one / z;
            if (hx < 0) {
                if (((ix - 0x3ff00000) | yisint) == 0) z = (z - z) / (z - z);
                else if (yisint == 1) z = -z;
            }
            return z;
        }
    }
    n = (int32_t)((uint32_t)hx >> 31) - 1;
    if ((n | yisint) == 0) return (x - x) / (x - x);
    s = one;
    if ((n | (yisint - 1)) == 0) s = -one;

    if (iy > 0x41e00000) {
        if (iy > 0x43f00000) {
            if (ix <= 0x3fefffff) return (hy < 0) ? huge * huge : tiny * tiny;
            if (ix >= 0x3ff00000) return (hy > 0) ? huge * huge : tiny * tiny;
        }
        if (ix < 0x3fefffff) return (hy < 0) ? s * huge * huge : s * tiny * tiny;
        if (ix > 0x3ff00000) return (hy > 0) ? s * huge * huge : s * tiny * tiny;
        t = ax - one;
        w = (t * t) * (0.5 - t * (0.3333333333333333333333 - t * 0.25));
        u = ivln2_h * t;
        v = t * ivln2_l - w * ivln2;
        t1 = u + v;
        t1 = lo_setlo(t1, 0);
        t2 = v - (t1 - u);
    } else {
        double ss, s2, s_h, s_l, t_h, t_l;
        n = 0;
        if (ix < 0x00100000) { ax *= two53; n -= 53; ix = (int32_t)lo_hi(ax); }
        n += ((ix) >> 20) - 0x3ff;
        j = ix & 0x000fffff;
        ix = j | 0x3ff00000;
        if (j <= 0x3988E) k = 0;
        else if (j < 0xBB67A) k = 1;
        else { k = 0; n += 1; ix -= 0x00100000; }
        ax = lo_sethi(ax, (uint32_t)ix);
        u = ax - bp[k];
        v = one / (ax + bp[k]);
        ss = u * v;
        s_h = ss;
        s_h = lo_setlo(s_h, 0);
        t_h = zero;
        t_h = lo_sethi(t_h, (uint32_t)(((ix >> 1) | 0x20000000) + 0x00080000 + (k << 18)));
        t_l = ax - (t_h - bp[k]);
        s_l = v * ((u - s_h * t_h) - s_h * t_l);
        s2 = ss * ss;
        r = s2 * s2 * (L1 + s2 * (L2 + s2 * (L3 + s2 * (L4 + s2 * (L5 + s2 * L6)))));
        r += s_l * (s_h + ss);
        s2 = s_h * s_h;
        t_h = 3.0 + s2 + r;
        t_h = lo_setlo(t_h, 0);
        t_l = r - ((t_h - 3.0) - s2);
        u = s_h * t_h;
        v = s_l * t_h + t_l * ss;
        p_h = u + v;
        p_h = lo_setlo(p_h, 0);
        p_l = v - (p_h - u);
        z_h = cp_h * p_h;
        z_l = cp_l * p_h + p_l * cp + dp_l[k];
        t = (double)n;
        t1 = (((z_h + z_l) + dp_h[k]) + t);
        t1 = lo_setlo(t1, 0);
        t2 = z_l - (((t1 - t) - dp_h[k]) - z_h);
    }
    y1 = y;
    y1 = lo_setlo(y1, 0);
    p_l = (y - y1) * t1 + y * t2;
    p_h = y1 * t1;
    z = p_l + p_h;
    j = (int32_t)lo_hi(z); i = (int32_t)lo_lo(z);
    if (j >= 0x40900000) {
        if (((j - 0x40900000) | i) != 0) return s * huge * huge;
        else { if (p_l + ovt > z - p_h) return s * huge * huge; }
    } else if ((j & 0x7fffffff) >= 0x4090cc00) {
        if (((j - (int32_t)0xc090cc00) | i) != 0) return s * tiny * tiny;
        else { if (p_l <= z - p_h) return s * tiny * tiny; }
    }
    i = j & 0x7fffffff;
    k = (i >> 20) - 0x3ff;
    n = 0;
    if (i > 0x3fe00000) {
        n = j + (0x00100000 >> (k + 1));
        k = ((n & 0x7fffffff) >> 20) - 0x3ff;
        t = zero;
        t = lo_sethi(t, (uint32_t)(n & ~(0x000fffff >> k)));
        n = ((n & 0x000fffff) | 0x00100000) >> (20 - k);
        if (j < 0) n = -n;
        p_h -= t;
    }
    t = p_l + p_h;
    t = lo_setlo(t, 0);
    u = t * lg2_h;
    v = (p_l - (t - p_h)) * lg2 + t * lg2_l;
    z = u + v;
    w = v - (z - u);
    t = z * z;
    t1 = z - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    r = (z * t1) / ((t1 - two) - (w + z * w)); /* sic: V8 7.8 groups the divisor this way (ieee754.cc) -- differs from Sun fdlibm */
    z = one - (r - z);
    j = (int32_t)lo_hi(z);
    j += (n << 20);
    if ((j >> 20) <= 0) z = v8_scalbn(z, n);
    else z = lo_sethi(z, (uint32_t)j);
    return s * z;
}

/* JavaScript ToInt32 (the `0 | x` idiom and Int32Array stores) */
static inline int32_t js_toint32(double d) {
    if (!(d == d) || isinf(d)) return 0;
    if (d >= -2147483648.0 && d <= 2147483647.0) return (int32_t)d;     /* truncation toward zero */
    double t = trunc(d);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

#endif
