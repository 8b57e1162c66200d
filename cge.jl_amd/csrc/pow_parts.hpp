// pow_parts.hpp -- (1 - D)^alpha as 2^(alpha * log2(1 - D)) with the logarithm stored once per score (src/divergence.jl:146,
// :430).  Shared by the power-matrix kernels (kernels_fit.hip) and the prologue of the persistent fit (kernels_fitp.hip):
// one definition, so both produce the same bits.
#pragma once
#include "common.hpp"

// ------------------------------------------------------------------------------------------------
// GD = (1 - D)^alpha, forty times per score on the same D.  x^alpha = 2^(alpha * log2 x), and log2(1 - D) does not depend
// on alpha: it is computed ONCE per score to ~70 bits (a double plus a float correction), and an alpha then costs one
// exp2 of a double-double exponent per element instead of a full pow (which spends most of its time on that logarithm).
// Accuracy: below one ulp (log2 to 2^-70, the product alpha*L exact through an fma residual, 2^f as 1 + f*ln2 + f^2*P(f)
// with the leading term carried as hi + lo) -- the same class as the library pow this replaces (option "pow_exp2" = 0).
struct dd_t { double h, l; };
__device__ __forceinline__ dd_t two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd_t quick_two_sum(double a, double b) { // |a| >= |b|
    const double s = a + b;
    return {s, b - (s - a)};
}
__device__ __forceinline__ dd_t dd_add(dd_t x, dd_t y) {
    const dd_t s = two_sum(x.h, y.h);
    return quick_two_sum(s.h, s.l + (x.l + y.l));
}
__device__ __forceinline__ dd_t dd_mul(dd_t x, dd_t y) {
    const double p = x.h * y.h;
    const double e = fma(x.h, y.h, -p) + (x.h * y.l + x.l * y.h);
    return quick_two_sum(p, e);
}
__device__ __forceinline__ dd_t dd_div(dd_t x, dd_t y) { // three quotient digits
    const double q1 = x.h / y.h;
    dd_t r = dd_add(x, dd_mul(y, {-q1, 0.0}));
    const double q2 = r.h / y.h;
    r = dd_add(r, dd_mul(y, {-q2, 0.0}));
    const double q3 = r.h / y.h;
    const dd_t q = quick_two_sum(q1, q2);
    return dd_add(q, {q3, 0.0});
}
// log2(x), x > 0 finite and normal, as hi + lo (|lo| <= ulp(hi)/2): x = m*2^e with m in [sqrt(1/2), sqrt(2)),
// log2 m = (2/ln 2) * atanh(s), s = (m-1)/(m+1), atanh(s) = s*(1 + z*(1/3 + z*(1/5 + z*R(z)))), z = s^2 <= 0.0295;
// R in double (it enters below 2^-13 of the result), the rest in double-double.
__device__ __forceinline__ dd_t log2_dd(double x) {
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
    const dd_t s = dd_div({m - 1.0, 0.0}, two_sum(m, 1.0));
    const dd_t z = dd_mul(s, s);
    double R = 1.0 / 33.0;
#pragma unroll
    for (int k = 31; k >= 7; k -= 2) R = fma(R, z.h, 1.0 / (double)k);
    const dd_t A = dd_add({0.20000000000000001, -1.1102230246251566e-17}, dd_mul(z, {R, 0.0}));
    const dd_t B = dd_add({0.33333333333333331, 1.8503717077085941e-17}, dd_mul(z, A));
    const dd_t C = dd_add({1.0, 0.0}, dd_mul(z, B));
    const dd_t T = dd_mul(s, C);
    const dd_t L = dd_mul(T, {2.8853900817779268, 4.0710547481862066e-17});
    return dd_add({(double)e, 0.0}, L);
}
// the two stored parts of log2(1 - D): 1 - D == 0 -> -inf (the power is 0), NaN stays NaN
__device__ __forceinline__ void log_parts(double d, double &Lh, float &Ll) {
    const double x = 1.0 - d;
    if (x > 0.0 && x < 1.7976931348623157e308) {
        const dd_t L = log2_dd(x);
        Lh = L.h;
        Ll = (float)L.l;
    } else {
        Lh = (x == 0.0) ? -__builtin_huge_val() : ((x != x) ? x : log2(x)); // 0, NaN, (never: negative / inf)
        Ll = 0.0f;
    }
}
// 2^(alpha * (Lh + Ll))
__device__ __forceinline__ double exp2_parts(double alpha, double Lh, float Ll) {
    const double p = alpha * Lh;
    // NaN stays NaN; -inf and anything below the subnormals is 0.  Branch-free (the select is at the end; what the arithmetic
    // below makes of such a p is discarded): sixty-four of these per lane sit in the prologue of the persistent fit, where
    // divergent branches would cost it scalar registers it does not have
    const bool tiny = !(p > -1100.0);
    const double yl = fma(alpha, Lh, -p) + alpha * (double)Ll;
    const double k = rint(p);
    const double f = (p - k) + yl; // |f| <= 1/2 (+ a rounding)
    const double LN2H = 0.69314718055994529, LN2L = 2.3190468138462996e-17;
    const double t1 = f * LN2H;
    const double t1e = fma(f, LN2H, -t1) + f * LN2L;
    double P = 6.7787263548225451e-14; // ln2^i / i!, i = 14 .. 2
    P = fma(P, f, 1.3691488853904128e-12);
    P = fma(P, f, 2.5678435993488206e-11);
    P = fma(P, f, 4.4455382718708116e-10);
    P = fma(P, f, 7.0549116208011234e-09);
    P = fma(P, f, 1.01780860092397e-07);
    P = fma(P, f, 1.321548679014431e-06);
    P = fma(P, f, 1.5252733804059841e-05);
    P = fma(P, f, 0.00015403530393381609);
    P = fma(P, f, 0.0013333558146428443);
    P = fma(P, f, 0.0096181291076284769);
    P = fma(P, f, 0.055504108664821583);
    P = fma(P, f, 0.24022650695910072);
    const double q = t1 + fma(f * f, P, t1e);
    const double r = ldexp(1.0 + q, tiny ? 0 : (int)k);
    return tiny ? ((p != p) ? p : 0.0) : r;
}
