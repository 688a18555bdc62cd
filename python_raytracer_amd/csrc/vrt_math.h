// vrt_math.h -- portable, deterministic binary64 sin / cos / pow for the trace path.
//
// Why this exists: the reference computes its ray directions and absorption falloff
// with CPython's math.sin / math.cos / float ** float (reference lib.py:323-338, 361-376,
// 450, 465), i.e. the host libm.  A GPU has no glibc, and two different "<1 ulp" libms do
// not agree bit for bit.  These routines evaluate the functions in double-double
// arithmetic (~100 significant bits) and round once, so the result is the correctly
// rounded value except within ~2^-45 ulp of a rounding boundary -- the same value a
// correctly rounded libm returns, on the host and on gfx950 alike (only +,-,*,/ and fma,
// all IEEE-exact; build with -ffp-contract=off).
//
// The same header is compiled by hipcc into the kernels and by gcc into the test oracle's
// "portable" libm mode; the oracle's other mode calls glibc and is what pins the oracle to
// the reference (tests/test_oracle_golden.py); tests/test_math.py checks these functions against
// the correctly rounded value (mpmath, 300 bits) and counts how often glibc differs from it
// (see DESIGN.md "Arithmetic").
//
// Domain: vrt_sin/vrt_cos |x| <= 2^20 (camera half-angles are < 2 rad); vrt_pow x > 0,
// finite y, result inside the normal range.  Outside the domain the functions return NaN.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VRT_HD __host__ __device__ static inline
#define VRT_CONST static __device__ const
#else
#define VRT_HD static inline
#define VRT_CONST static const
#endif

#include "vrt_math_consts.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

typedef struct { double h, l; } vrt_dd;

VRT_HD vrt_dd vrt_dd_make(double h, double l) { vrt_dd r; r.h = h; r.l = l; return r; }

// error-free transforms
VRT_HD vrt_dd vrt_two_sum(double a, double b) {
    double s = a + b;
    double bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return vrt_dd_make(s, e);
}
VRT_HD vrt_dd vrt_fast_two_sum(double a, double b) {  // |a| >= |b| or a == 0
    double s = a + b;
    double e = b - (s - a);
    return vrt_dd_make(s, e);
}
VRT_HD vrt_dd vrt_two_prod(double a, double b) {
    double p = a * b;
    double e = __builtin_fma(a, b, -p);
    return vrt_dd_make(p, e);
}

VRT_HD vrt_dd vrt_dd_add(vrt_dd a, vrt_dd b) {
    vrt_dd s = vrt_two_sum(a.h, b.h);
    vrt_dd t = vrt_two_sum(a.l, b.l);
    s.l += t.h;
    s = vrt_fast_two_sum(s.h, s.l);
    s.l += t.l;
    return vrt_fast_two_sum(s.h, s.l);
}
VRT_HD vrt_dd vrt_dd_add_d(vrt_dd a, double b) {
    vrt_dd s = vrt_two_sum(a.h, b);
    s.l += a.l;
    return vrt_fast_two_sum(s.h, s.l);
}
// a + b for |a.h| >= |b.h| with no heavy cancellation (Horner steps whose coefficient dominates): 8 flops
VRT_HD vrt_dd vrt_dd_add_ord(vrt_dd a, vrt_dd b) {
    vrt_dd s = vrt_fast_two_sum(a.h, b.h);
    s.l += a.l + b.l;
    return vrt_fast_two_sum(s.h, s.l);
}
VRT_HD vrt_dd vrt_dd_neg(vrt_dd a) { return vrt_dd_make(-a.h, -a.l); }
VRT_HD vrt_dd vrt_dd_mul(vrt_dd a, vrt_dd b) {
    vrt_dd p = vrt_two_prod(a.h, b.h);
    p.l += a.h * b.l;
    p.l += a.l * b.h;
    return vrt_fast_two_sum(p.h, p.l);
}
VRT_HD vrt_dd vrt_dd_mul_d(vrt_dd a, double b) {
    vrt_dd p = vrt_two_prod(a.h, b);
    p.l += a.l * b;
    return vrt_fast_two_sum(p.h, p.l);
}
VRT_HD vrt_dd vrt_dd_div(vrt_dd a, vrt_dd b) {
    double q1 = a.h / b.h;
    vrt_dd r = vrt_dd_add(a, vrt_dd_neg(vrt_dd_mul_d(b, q1)));
    double q2 = r.h / b.h;
    r = vrt_dd_add(r, vrt_dd_neg(vrt_dd_mul_d(b, q2)));
    double q3 = r.h / b.h;
    vrt_dd q = vrt_fast_two_sum(q1, q2);
    return vrt_dd_add_d(q, q3);
}

// ---------------------------------------------------------------------------------
// sin / cos
// ---------------------------------------------------------------------------------
// r = x - k*pi/2 in double-double, k = nearest integer to x*2/pi; returns k & 3 in *quad.
VRT_HD vrt_dd vrt_reduce_pio2(double x, int* quad) {
    double kf = __builtin_rint(x * VRT_2OPI);
    *quad = (int)((long long)kf & 3);
    if (kf == 0.0) return vrt_dd_make(x, 0.0);
    // x - kf*P0 is exact (|x| >= pi/4 here, both are multiples of 2^-53 and the result is < 1)
    double t = __builtin_fma(-kf, VRT_PIO2[0], x);
    vrt_dd r = vrt_dd_make(t, 0.0);
    r = vrt_dd_add(r, vrt_dd_neg(vrt_two_prod(kf, VRT_PIO2[1])));
    r = vrt_dd_add(r, vrt_dd_neg(vrt_two_prod(kf, VRT_PIO2[2])));
    r = vrt_dd_add(r, vrt_dd_neg(vrt_two_prod(kf, VRT_PIO2[3])));
    return r;
}

// Taylor kernels on |r| <= pi/4 (+ a hair): 15 terms each, all in double-double.
VRT_HD vrt_dd vrt_sin_kernel(vrt_dd r) {
    vrt_dd z = vrt_dd_mul(r, r);
    // sum_{n=0..14} (-1)^n z^n / (2n+1)!   by Horner from the top
    vrt_dd p = vrt_dd_make(VRT_INVFACT[29][0], VRT_INVFACT[29][1]);
    for (int n = 13; n >= 0; --n) {
        p = vrt_dd_mul(p, z);
        vrt_dd c = vrt_dd_make(VRT_INVFACT[2 * n + 1][0], VRT_INVFACT[2 * n + 1][1]);
        p = vrt_dd_add(c, vrt_dd_neg(p));
    }
    return vrt_dd_mul(p, r);
}
VRT_HD vrt_dd vrt_cos_kernel(vrt_dd r) {
    vrt_dd z = vrt_dd_mul(r, r);
    vrt_dd p = vrt_dd_make(VRT_INVFACT[28][0], VRT_INVFACT[28][1]);
    for (int n = 13; n >= 0; --n) {
        p = vrt_dd_mul(p, z);
        vrt_dd c = vrt_dd_make(VRT_INVFACT[2 * n][0], VRT_INVFACT[2 * n][1]);
        p = vrt_dd_add(c, vrt_dd_neg(p));
    }
    return p;
}

// Fast kernels: the four leading Taylor terms in double-double, the tail in plain binary64.  Relative error
// < 2^-70 (the binary64 tail enters at z^4*2.8e-6 resp. z^5*2.8e-7 of the result); the callers accept the
// result only if rounding is unambiguous under a 2^-70 bound (Ziv's test), else they use the full kernels.
// (z = r * r: a caller that wants sin and cos of the same r squares it once)
VRT_HD vrt_dd vrt_sin_fast_z(vrt_dd r, vrt_dd z) {
    const double zh = z.h;
    double t = VRT_INVFACT[27][0];  // sum_{n>=4} (-1)^n z^(n-4) / (2n+1)!
    t = VRT_INVFACT[25][0] - t * zh;
    t = VRT_INVFACT[23][0] - t * zh;
    t = VRT_INVFACT[21][0] - t * zh;
    t = VRT_INVFACT[19][0] - t * zh;
    t = VRT_INVFACT[17][0] - t * zh;
    t = VRT_INVFACT[15][0] - t * zh;
    t = VRT_INVFACT[13][0] - t * zh;
    t = VRT_INVFACT[11][0] - t * zh;
    // c9 - z*t with c9 in double-double
    vrt_dd p = vrt_dd_add_ord(vrt_dd_make(VRT_INVFACT[9][0], VRT_INVFACT[9][1]), vrt_dd_neg(vrt_dd_mul_d(z, t)));
    for (int n = 3; n >= 0; --n) {
        p = vrt_dd_mul(p, z);
        p = vrt_dd_add_ord(vrt_dd_make(VRT_INVFACT[2 * n + 1][0], VRT_INVFACT[2 * n + 1][1]), vrt_dd_neg(p));
    }
    return vrt_dd_mul(p, r);
}
VRT_HD vrt_dd vrt_sin_fast(vrt_dd r) { return vrt_sin_fast_z(r, vrt_dd_mul(r, r)); }
VRT_HD vrt_dd vrt_cos_fast_z(vrt_dd z) {
    const double zh = z.h;
    double t = VRT_INVFACT[28][0];  // sum_{n>=5} (-1)^(n-5) z^(n-5) / (2n)!
    t = VRT_INVFACT[26][0] - t * zh;
    t = VRT_INVFACT[24][0] - t * zh;
    t = VRT_INVFACT[22][0] - t * zh;
    t = VRT_INVFACT[20][0] - t * zh;
    t = VRT_INVFACT[18][0] - t * zh;
    t = VRT_INVFACT[16][0] - t * zh;
    t = VRT_INVFACT[14][0] - t * zh;
    t = VRT_INVFACT[12][0] - t * zh;
    t = VRT_INVFACT[10][0] - t * zh;
    // c8 - z*t
    vrt_dd p = vrt_dd_add_ord(vrt_dd_make(VRT_INVFACT[8][0], VRT_INVFACT[8][1]), vrt_dd_neg(vrt_dd_mul_d(z, t)));
    for (int n = 3; n >= 0; --n) {
        p = vrt_dd_mul(p, z);
        p = vrt_dd_add_ord(vrt_dd_make(VRT_INVFACT[2 * n][0], VRT_INVFACT[2 * n][1]), vrt_dd_neg(p));
    }
    return p;
}
VRT_HD vrt_dd vrt_cos_fast(vrt_dd r) { return vrt_cos_fast_z(vrt_dd_mul(r, r)); }
// Ziv rounding test: v is within 2^-70 |v| of the true value; returns 1 and the rounded value if unambiguous
VRT_HD int vrt_round_test(vrt_dd v, double* out) {
    const double e = __builtin_fabs(v.h) * 0x1p-70;
    const double a = v.h + (v.l - e), b = v.h + (v.l + e);
    *out = a;
    return a == b;
}

VRT_HD double vrt_sin(double x) {
    if (x == 0.0) return x;
    if (!(__builtin_fabs(x) <= 1048576.0)) return __builtin_nan("");
    int q;
    vrt_dd r = vrt_reduce_pio2(x, &q);
    double out;
    vrt_dd v = (q & 1) ? vrt_cos_fast(r) : vrt_sin_fast(r);
    if (!vrt_round_test(v, &out)) {
        v = (q & 1) ? vrt_cos_kernel(r) : vrt_sin_kernel(r);
        out = v.h;
    }
    return (q & 2) ? -out : out;
}
VRT_HD double vrt_cos(double x) {
    if (!(__builtin_fabs(x) <= 1048576.0)) return __builtin_nan("");
    int q;
    vrt_dd r = vrt_reduce_pio2(x, &q);
    double out;
    vrt_dd v = (q & 1) ? vrt_sin_fast(r) : vrt_cos_fast(r);
    if (!vrt_round_test(v, &out)) {
        v = (q & 1) ? vrt_sin_kernel(r) : vrt_cos_kernel(r);
        out = v.h;
    }
    return ((q + 1) & 2) ? -out : out;
}

// out = { sin a, cos a, sin b, cos b }, bit for bit what vrt_sin / vrt_cos return (the lens quaternion's four values,
// lib.py:323-338).  When both arguments lie inside (-pi/4, pi/4) and neither is zero -- every camera lens -- the four
// fast kernels run as one straight-line block: the same operations on the same values as the single calls make, but
// four independent chains side by side instead of four calls with a branch each, and r * r once per argument.
// Returns 0 if that block does not apply or one of its results fails the rounding test: the caller then uses
// vrt_sin / vrt_cos (vrt_sincos2 below does).
VRT_HD int vrt_sincos2_fast(double a, double b, double out[4]) {
    const double ka = __builtin_rint(a * VRT_2OPI), kb = __builtin_rint(b * VRT_2OPI);  // (vrt_reduce_pio2's k)
    if (!(ka == 0.0 && kb == 0.0 && a != 0.0 && b != 0.0)) return 0;
    const vrt_dd ra = vrt_dd_make(a, 0.0), rb = vrt_dd_make(b, 0.0);
    const vrt_dd za = vrt_dd_mul(ra, ra), zb = vrt_dd_mul(rb, rb);
    const vrt_dd sa = vrt_sin_fast_z(ra, za), ca = vrt_cos_fast_z(za);
    const vrt_dd sb = vrt_sin_fast_z(rb, zb), cb = vrt_cos_fast_z(zb);
    const int ok = vrt_round_test(sa, &out[0]) & vrt_round_test(ca, &out[1]) & vrt_round_test(sb, &out[2]) &
                   vrt_round_test(cb, &out[3]);
    return ok;
}
VRT_HD void vrt_sincos2(double a, double b, double out[4]) {
    if (vrt_sincos2_fast(a, b, out)) return;
    out[0] = vrt_sin(a);
    out[1] = vrt_cos(a);
    out[2] = vrt_sin(b);
    out[3] = vrt_cos(b);
}

// ---------------------------------------------------------------------------------
// pow(x, y) for x > 0
// ---------------------------------------------------------------------------------
// log(x) = e*ln2 + 2*atanh((m-1)/(m+1)), m in [sqrt(1/2), sqrt(2))
VRT_HD vrt_dd vrt_log_dd(double x) {
    union { double d; uint64_t u; } cv;
    cv.d = x;
    int e = (int)((cv.u >> 52) & 0x7ff) - 1023;
    cv.u = (cv.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;  // m in [1,2)
    double m = cv.d;
    if (m >= 1.4142135623730951) { m *= 0.5; e += 1; }
    vrt_dd num = vrt_dd_make(m - 1.0, 0.0);  // exact
    vrt_dd den = vrt_two_sum(m, 1.0);
    vrt_dd s = vrt_dd_div(num, den);
    vrt_dd z = vrt_dd_mul(s, s);
    // |s| <= 0.1716: z^n/(2n+1) < 2^-110 for n >= 22
    vrt_dd p = vrt_dd_make(VRT_INVODD[22][0], VRT_INVODD[22][1]);
    for (int n = 21; n >= 0; --n) {
        p = vrt_dd_mul(p, z);
        p = vrt_dd_add(vrt_dd_make(VRT_INVODD[n][0], VRT_INVODD[n][1]), p);
    }
    p = vrt_dd_mul(p, s);
    p.h *= 2.0;
    p.l *= 2.0;
    if (e != 0) {
        double ef = (double)e;
        // e*ln2 with a 3-part ln2: first product exact (ln2[0] has trailing zeros? not relied on -> two_prod)
        vrt_dd a = vrt_two_prod(ef, VRT_LN2[0]);
        vrt_dd b = vrt_two_prod(ef, VRT_LN2[1]);
        vrt_dd c = vrt_dd_add(a, b);
        c = vrt_dd_add_d(c, ef * VRT_LN2[2]);
        p = vrt_dd_add(c, p);
    }
    return p;
}

// exp of a double-double, result as double-double times 2^k
VRT_HD double vrt_exp_dd_round(vrt_dd a) {
    double kf = __builtin_rint(a.h * VRT_INVLN2);
    vrt_dd r = a;
    if (kf != 0.0) {
        r = vrt_dd_add(r, vrt_dd_neg(vrt_two_prod(kf, VRT_LN2[0])));
        r = vrt_dd_add(r, vrt_dd_neg(vrt_two_prod(kf, VRT_LN2[1])));
        r = vrt_dd_add_d(r, -(kf * VRT_LN2[2]));
    }
    // |r| <= 0.3466: r^n/n! < 2^-109 for n >= 24
    vrt_dd p = vrt_dd_make(VRT_INVFACT[25][0], VRT_INVFACT[25][1]);
    for (int n = 24; n >= 0; --n) {
        p = vrt_dd_mul(p, r);
        p = vrt_dd_add(vrt_dd_make(VRT_INVFACT[n][0], VRT_INVFACT[n][1]), p);
    }
    int k = (int)kf;
    if (k < -1000 || k > 1000) return __builtin_nan("");
    union { double d; uint64_t u; } sc;
    sc.u = (uint64_t)(k + 1023) << 52;
    return p.h * sc.d;  // exact scaling inside the normal range
}

VRT_HD double vrt_pow(double x, double y) {
    if (y == 0.0 || x == 1.0) return 1.0;
    if (!(x > 0.0) || !(x < 1.0e300) || !(__builtin_fabs(y) < 1.0e300) || x < 1.0e-300) return __builtin_nan("");
    vrt_dd l = vrt_log_dd(x);
    vrt_dd t = vrt_dd_mul_d(l, y);
    return vrt_exp_dd_round(t);
}
