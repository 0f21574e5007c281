// Lean fp64 elementary functions for the wave kernel (gfx950).  They exist because the fast path is
// VALU-issue bound: the stock device-library calls carry long constant pools and special-case
// code that cost registers and instructions on every row.  Accuracy is ~1-2 ulp on the domains
// stated below -- far inside the 1e-9 the parity tests hold the engine to.
#pragma once

#include "pla_device.h"

namespace pla {

// ---- range reduction x = k*ln2/256 + r shared by every exponential -----------------------------
constexpr int kTabN = 256;
constexpr double kC256 = 369.32993046757463;              // 256 / ln 2
constexpr double kMagic = 6755399441055744.0;             // 1.5 * 2^52: round-to-nearest-int trick
constexpr double kLn2Hi256 = 2.70760617331688990816e-03;  // 6.93147180369123816490e-01 / 256
constexpr double kLn2Lo256 = 7.45396456746323320e-13;     // 1.90821492927058770002e-10 / 256
constexpr double kLn2_256 = 2.70760617406228627432e-03;    // ln 2 / 256 in one piece (|k| < 2^18: error < 1e-13)

struct Red256 {
  int k;     // round(x * 256 / ln 2)
  double r;  // x - k ln2/256, |r| <= 0.00136
};
__device__ __forceinline__ Red256 reduce256(double x) {
  const double t = fma(x, kC256, kMagic);
  Red256 o;
  o.k = (int)(unsigned)(__double_as_longlong(t) & 0xffffffffll);  // low mantissa bits hold k
  const double kf = t - kMagic;
  const double r = fma(kf, -kLn2Hi256, x);
  o.r = fma(kf, -kLn2Lo256, r);
  return o;
}
__device__ __forceinline__ int key256(double x) {
  const double t = fma(x, kC256, kMagic);
  return (int)(unsigned)(__double_as_longlong(t) & 0xffffffffll);
}
__device__ __forceinline__ double scale_exp(double v, int es) {  // v * 2^(es >> 20); result must stay normal
  long long b = __double_as_longlong(v);
  b += (long long)es << 32;
  return __longlong_as_double(b);
}
// The table: entry j (16 bytes) = {2^(j/256), 2^(-j/256)} with the high words biased by -/+ (j << 12).
// With k = 256 e + j the scaled entry 2^e * 2^(j/256) is then  hi + (k << 12)  -- one integer
// multiply-add, no masking of k -- and likewise  hi - (k << 12)  for 2^-e * 2^(-j/256).
__device__ __forceinline__ void exp_table_entry(double* tab, int j) {
  const double p = exp2((double)j * (1.0 / kTabN)), n = exp2(-(double)j * (1.0 / kTabN));
  tab[2 * j] = __hiloint2double(__double2hiint(p) - (j << 12), __double2loint(p));
  tab[2 * j + 1] = __hiloint2double(__double2hiint(n) + (j << 12), __double2loint(n));
}
// e^x for x in [-700, 709]; results must stay normal
__device__ __forceinline__ double exp_tab(double x, const double* tab) {
  const Red256 q = reduce256(x);
  const double tj = tab[2 * (q.k & 255)];
  const double r2 = q.r * q.r;
  const double E = fma(fma(4.16666666666666666667e-02, r2, 0.5), r2, 1.0);
  const double O = fma(1.66666666666666666667e-01, r2, 1.0);
  return __hiloint2double(__double2hiint(tj) + (q.k << 12), __double2loint(tj)) * fma(q.r, O, E);
}
// e^x and e^-x from one range reduction, x in [-700, 700]
__device__ __forceinline__ void exp_pair(double x, const double* tab, double& ep, double& en) {
  const Red256 q = reduce256(x);
  const double2 tj = *reinterpret_cast<const double2*>(tab + 2 * (q.k & 255));
  const double r2 = q.r * q.r;
  const double E = fma(fma(4.16666666666666666667e-02, r2, 0.5), r2, 1.0);
  const double O = fma(1.66666666666666666667e-01, r2, 1.0);
  ep = __hiloint2double(__double2hiint(tj.x) + (q.k << 12), __double2loint(tj.x)) * fma(q.r, O, E);
  en = __hiloint2double(__double2hiint(tj.y) - (q.k << 12), __double2loint(tj.y)) * fma(-q.r, O, E);
}
// e^x for any x <= 0 (and NaN -> caller's problem): clamps where e^x underflows anyway
__device__ __forceinline__ double exp_neg(double x, const double* tab) { return exp_tab(fmax(x, -700.0), tab); }

// e^z - 1.  |z| < 2^-5: Taylor to z^9 (no cancellation); otherwise e^z - 1 from the table (the
// subtraction loses at most 5 bits there).  z > 709 -> +inf, z < -700 -> -1.  NaN -> NaN.
__device__ __forceinline__ double expm1_tab(double z, const double* tab) {
  const double zc = fmin(fmax(z, -700.0), 709.0);
  double res = exp_tab(zc, tab) - 1.0;
  const bool small = fabs(zc) < 0.03125;
  if (__builtin_amdgcn_ballot_w64(small) != 0ull) {  // wave-uniform: most calls have no small argument
    double p = 2.75573192239858906526e-06;  // 1/9!
    p = fma(p, zc, 2.48015873015873015873e-05);
    p = fma(p, zc, 1.98412698412698412698e-04);
    p = fma(p, zc, 1.38888888888888888889e-03);
    p = fma(p, zc, 8.33333333333333333333e-03);
    p = fma(p, zc, 4.16666666666666666667e-02);
    p = fma(p, zc, 1.66666666666666666667e-01);
    p = fma(p, zc, 0.5);
    p = fma(p * zc, zc, zc);
    if (small) res = p;
  }
  if (z > 709.0) res = pinf();
  if (z != z) res = z;
  return res;
}

// 1/d and a/d to ~1 ulp without the full IEEE division sequence (d normal, no overflow games)
__device__ __forceinline__ double recip_fast(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
// v_rcp_f64 is good to 2^-23; one Newton step squares that (2^-46) and the residual correction of the quotient squares it
// again: q (1 - eps^2) with eps = 2^-46 -- the second Newton step of recip_fast would be wasted here
#ifndef PLA_DIV_ONE_NEWTON
#define PLA_DIV_ONE_NEWTON 1
#endif
__device__ __forceinline__ double div_fast(double a, double d) {
#if PLA_DIV_ONE_NEWTON
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
#else
  const double r = recip_fast(d);
#endif
  const double q = a * r;
  return fma(fma(-q, d, a), r, q);
}

// natural log (fdlibm e_log.c kernel, <1 ulp); zero, negative, NaN, inf and subnormal inputs are
// patched up afterwards so the IEEE results match log().  Only used off the hot path now.
__device__ __forceinline__ double log_fast(double x) {
  const bool sub = (x < 2.2250738585072014e-308) && (x > 0.0);
  const double xs = sub ? x * 18014398509481984.0 : x;  // 2^54
  int e = __builtin_amdgcn_frexp_exp(xs);               // xs = m * 2^e, m in [0.5, 1)
  double m = __builtin_amdgcn_frexp_mant(xs);
  if (m < 0.70710678118654752440) { m += m; e -= 1; }
  if (sub) e -= 54;
  const double f = m - 1.0;
  const double s = div_fast(f, 2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  double res = dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
  if (x == 0.0) res = -pinf();
  if (x < 0.0) res = qnan();
  if (x != x) res = x;
  if (x == pinf()) res = x;
  return res;
}

// ---- table-driven natural log --------------------------------------------------------------------
// x = 2^e * m with m in [sqrt(1/2), sqrt(2)) (so that x ~ 1 gives e = 0 and no cancellation between
// e ln2 and log m).  The low exponent bit and the top 7 mantissa bits of m pick a cell with centre c_j,
// |m/c_j - 1| < 2^-7; log m = log c_j + log1p(m/c_j - 1), the latter a degree-7 polynomial.  The two
// cells that touch m = 1 use c = 1 exactly, so log x keeps full RELATIVE accuracy as x -> 1.
// Entry j (16 bytes): {1/c_j, log c_j}; j < 128: binade [1/2, 1), j >= 128: binade [1, 2) (all cells
// are valid, so a mantissa that lands just outside [sqrt(1/2), sqrt(2)) by rounding is harmless).
constexpr int kLogTabN = 256;
__device__ __forceinline__ void log_table_entry(double* lt, int j) {
  double c = (j < 128) ? 0.5 * (1.0 + ((double)j + 0.5) * (1.0 / 128.0)) : 1.0 + ((double)(j - 128) + 0.5) * (1.0 / 128.0);
  if (j == 127 || j == 128) c = 1.0;  // the cells adjacent to m = 1
  lt[2 * j] = 1.0 / c;
  lt[2 * j + 1] = log(c);
}
// positive normal x: ~1 ulp of max(|log x|, 2^-52 |log x|...) -- absolute error < 2.5e-16 + 1.2e-16 |log x|.
// Everything else (0, negatives, NaN, inf, subnormals) takes the wave-uniform slow branch.
__device__ __forceinline__ double log_tab(double x, const double* lt) {
  // e = exponent of x * sqrt(2) - 1  ->  m = x * 2^-e in [sqrt(1/2), sqrt(2))
  const int e = __builtin_amdgcn_frexp_exp(x * 1.41421356237309504880) - 1;
  const double m = __builtin_amdgcn_ldexp(x, -e);
  const int j = (__double2hiint(m) >> 13) & (kLogTabN - 1);  // exponent lsb | mantissa[51:45]
  const double2 cj = *reinterpret_cast<const double2*>(lt + 2 * j);
  const double r = fma(m, cj.x, -1.0);  // |r| < 2^-7, exact cancellation of the leading bits
  double q = fma(r, 1.42857142857142857143e-01, -1.66666666666666666667e-01);
  q = fma(q, r, 0.2);
  q = fma(q, r, -0.25);
  q = fma(q, r, 3.33333333333333333333e-01);
  q = fma(q, r, -0.5);
  const double l1p = fma(r * r, q, r);
  double res = fma((double)e, 6.93147180559945309417e-01, cj.y) + l1p;
  const bool ok = __builtin_amdgcn_class(x, 0x100);  // positive normal
  if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) {
    const double alt = log_fast(x);
    if (!ok) res = alt;
  }
  return res;
}

}  // namespace pla
