// PSIS-weighted expectations and their function-specific Pareto k (reference: pyloo e_loo.py:56-264, 328-390,
// 430-465, 518-531, 557-559).  Per observation, from the same-shape matrices x (draws to average),
// lw (log-weights, any normalisation) and lr (raw log ratios for the diagnostics, = lw when not given):
//
//   w_s     = exp(lw_s - logsumexp(lw))                                   e_loo.py:557-559, 434
//   mean    = sum_s w_s x_s                                               437
//   var     = max((sum w x^2 - mean^2) / (1 - sum w^2), 0), 0 when x is allclose to x[0] or sum w^2 isclose to 1     518-531
//   k       = k_hat(h, lr) with h = x (mean), x^2 (variance / sd), None (quantiles)                                   226-236
//
// k_hat (328-390) as the reference EVALUATES it.  Its three tails -- the 20 largest r = exp(lr - max lr), the 20 smallest
// and the 20 largest h r -- go to _gpdfit as `tail - tail[-1]`: DESCENDING, last element exactly 0, where _gpdfit
// (psis.py:163-208) wants ascending values.  1 / ary[-1] is then +-inf (psis.py:188), every profile-likelihood weight NaN,
// none passes `w >= 10 eps` (psis.py:194-197), b_post = sum of nothing = 0, k_post = mean log1p(-0 ary) = 0 and
//       k = (n k_post + 5) / (n + 10) = 5 / (n + 10)        (1/6 for n = 20; NaN when the tail itself holds a NaN),
// whatever the draws are.  The value of k_hat is therefore decided by its guards alone, and those are what this kernel
// computes: tails shorter than 5 or allclose to their first element (+inf for r, -inf for h r: 353-354, 373-383), h allclose
// to h[0], exactly two distinct values in h, NaN / inf in h (359-366: k of r alone), and the NaN rules of Python's max
// (385-390).  `allclose(tail, tail[0])` over the n largest values is a COUNT: at least n values within
// atol + rtol |extreme| of the extreme -- no selection or sort is needed.
//
// One 256-thread workgroup per observation, three passes over the row (the second and third hit L2 / the Infinity
// Cache): (1) maxima of lw and lr, the statistics of h; (2) the weighted sums, r, the extremes of h r; (3) the counts.
// Any strides; f32 input is computed in f64 (the parity target is the reference on the f64-upcast data, as everywhere).
#pragma once

#include "pla_math.h"
#include "pla_wave.h"

namespace pla {

struct ELooParams {
  const void* x;
  const void* lw;
  const void* lr;  // == lw when the caller has no raw ratios (e_loo.py:223-224)
  int64_t n_obs;
  int n_draws;
  int64_t stride_obs, stride_draw;  // elements; the three matrices share shape and strides
  int tail_len;                     // 20 (e_loo.py:269)
  double* mean;    // [n_obs] or null
  double* var;     // [n_obs] or null
  double* k_mean;  // [n_obs] or null: k_hat(x, lr)
  double* k_var;   // [n_obs] or null: k_hat(x^2, lr)
  double* k_none;  // [n_obs] or null: k_hat(None, lr)
  // fast path (e_loo_wave_kernel) -> general kernel hand-over: rows the former declines, and how many
  unsigned* slow_list;
  unsigned long long* slow_count;
};

constexpr double kCloseRtol = 1e-5, kCloseAtol = 1e-8;  // np.allclose / np.isclose defaults

// np.sort puts NaN last, so the n "largest" (or smallest) hold a NaN only when fewer than n values are not NaN
__device__ __forceinline__ double tail_piece(int n_tail, double n_valid, double n_close, double special) {
  if (n_tail < 5) return special;                       // e_loo.py:353, 373, 379
  if (n_valid < (double)n_tail) return qnan();          // a NaN inside the tail: allclose is False and _gpdfit returns NaN
  if (n_close >= (double)n_tail) return special;        // allclose(tail, tail[0])
  return 5.0 / ((double)n_tail + 10.0);                 // the degenerate _gpdfit (see the header)
}

// Python's max(a, b): a unless b > a
__device__ __forceinline__ double py_max(double a, double b) { return (b > a) ? b : a; }

// K reductions for the price of one barrier pair: op[k] = 0 sum, 1 max, 2 min; every thread ends with all K results
template <int K, int BLOCK>
__device__ __forceinline__ void block_reduce_k(double (&v)[K], const int (&op)[K], double* lds) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (op[k] == 0) v[k] = wave_reduce<OpSum>(v[k]);
    else if (op[k] == 1) v[k] = wave_reduce<OpMax>(v[k]);
    else v[k] = wave_reduce<OpMin>(v[k]);
  }
  constexpr int NW = BLOCK / kWave;
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[w * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double r = lds[k];
#pragma unroll
    for (int i = 1; i < NW; ++i) {
      const double o = lds[i * K + k];
      r = op[k] == 0 ? r + o : (op[k] == 1 ? fmax(r, o) : fmin(r, o));
    }
    v[k] = r;
  }
}

// everything that follows the row's sums, extremes and counts (one lane): expectations (e_loo.py:437, 518-531) and k_hat's guards
__device__ __forceinline__ void e_loo_finish(const ELooParams& P, const int64_t r, const int n_tail, const unsigned flags, const double sa,
                                             const double sb, const double sc, const double sd, const double x0, const double q0,
                                             const double xdev, const double qdev, const double xmn, const double xmx, const double qmn,
                                             const double qmx, const bool two_x, const bool two_q, const double n_valid,
                                             const double n_close_r, const double c1l, const double c1r, const double c2l,
                                             const double c2r) {
  // ---- expectations ----
  const double mean = sb / sa, msq = sc / sa, wss = sd / (sa * sa);
  if (P.mean) P.mean[r] = mean;
  if (P.var) {
    double v;
    if (xdev <= kCloseAtol + kCloseRtol * fabs(x0) && !(flags & 4u)) v = 0.0;      // e_loo.py:520-521
    else if (fabs(wss - 1.0) <= kCloseAtol + kCloseRtol * 1.0) v = 0.0;             // 523-525
    else {
      v = (msq - mean * mean) / (1.0 - wss);                                          // 527-530
      v = (0.0 > v) ? 0.0 : v;  // Python's max(var, 0.0): a NaN variance stays NaN    531
    }
    P.var[r] = v;
  }
  // ---- k_hat ----
  const double k_r = tail_piece(n_tail, n_valid, n_close_r, pinf());                 // 353-357
  const auto k_of = [&](bool skip, double n_left, double n_right) {
    if (skip) return k_r;                                                             // 359-366
    const double kl = tail_piece(n_tail, n_valid, n_left, -pinf());                  // 373-377
    const double kr = tail_piece(n_tail, n_valid, n_right, -pinf());                 // 379-383
    const double k_hr = py_max(kl, kr);                                               // 385
    if (k_hr != k_hr && k_r != k_r) return qnan();                                    // 387-388
    return py_max(k_hr, k_r);                                                         // 390
  };
  const bool skip_x = (xdev <= kCloseAtol + kCloseRtol * fabs(x0) && !(flags & 4u)) || (two_x && xmn != xmx && !(flags & 4u)) ||
                      (flags & (4u | 8u)) != 0u;
  const bool skip_q = (qdev <= kCloseAtol + kCloseRtol * fabs(q0) && !(flags & 4u)) || (two_q && qmn != qmx && !(flags & 4u)) ||
                      (flags & (4u | 16u)) != 0u;
  if (P.k_mean) P.k_mean[r] = k_of(skip_x, c1l, c1r);
  if (P.k_var) P.k_var[r] = k_of(skip_q, c2l, c2r);
  if (P.k_none) P.k_none[r] = k_r;
}

// LIST: the rows of P.slow_list (what the wave kernel below declined) instead of all of them
template <typename T, int BLOCK, bool LIST = false>
__global__ __launch_bounds__(BLOCK) void e_loo_rows_kernel(ELooParams P) {
  __shared__ double red[(BLOCK / kWave) * 12];
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  const int n_tail = S < P.tail_len ? S : P.tail_len;
  const bool own_lr = P.lr != P.lw;
  const int64_t n_rows = LIST ? (int64_t)*P.slow_count : P.n_obs;
  for (int64_t ri = blockIdx.x; ri < n_rows; ri += gridDim.x) {
    const int64_t r = LIST ? (int64_t)P.slow_list[ri] : ri;
    const T* xr = reinterpret_cast<const T*>(P.x) + r * P.stride_obs;
    const T* wr = reinterpret_cast<const T*>(P.lw) + r * P.stride_obs;
    const T* rr = reinterpret_cast<const T*>(P.lr) + r * P.stride_obs;
    // ---- pass 1: maxima of the log-weights / log ratios, statistics of x and x^2 ----------------------------------
    const double x0 = (double)xr[0], q0 = x0 * x0;
    double mlw = -pinf(), mlr = -pinf();
    double xmn = pinf(), xmx = -pinf(), qmx = -pinf(), qmn = pinf(), xdev = 0.0, qdev = 0.0;
    unsigned flags = 0;  // 1: NaN in lw, 2: NaN in lr, 4: NaN in x, 8: inf in x, 16: inf in x^2
    for (int s = tid; s < S; s += BLOCK) {
      const double a = (double)wr[(int64_t)s * P.stride_draw];
      const double b = own_lr ? (double)rr[(int64_t)s * P.stride_draw] : a;
      const double x = (double)xr[(int64_t)s * P.stride_draw];
      const double q = x * x;
      if (a != a) flags |= 1u;
      if (b != b) flags |= 2u;
      if (x != x) flags |= 4u;
      if (fabs(x) == pinf()) flags |= 8u;
      if (q == pinf()) flags |= 16u;
      mlw = fmax(mlw, a);
      mlr = fmax(mlr, b);
      xmn = fmin(xmn, x); xmx = fmax(xmx, x);
      qmn = fmin(qmn, q); qmx = fmax(qmx, q);
      xdev = fmax(xdev, fabs(x - x0));
      qdev = fmax(qdev, fabs(q - q0));
    }
    {
      double a[8] = {mlw, mlr, xmn, xmx, qmn, qmx, xdev, qdev};
      const int op[8] = {1, 1, 2, 1, 2, 1, 1, 1};
      block_reduce_k<8, BLOCK>(a, op, red);
      mlw = a[0]; mlr = a[1]; xmn = a[2]; xmx = a[3]; qmn = a[4]; qmx = a[5]; xdev = a[6]; qdev = a[7];
    }
    flags = block_or_bits<BLOCK>(flags, red);
    if (flags & 1u) mlw = qnan();  // np.max propagates NaN (utils.py:346, e_loo.py:350)
    if (flags & 2u) mlr = qnan();
    // ---- pass 2: weighted sums; r = exp(lr - max lr); extremes of x r and x^2 r; the two-distinct-values test ------
    double sa = 0.0, sb = 0.0, sc = 0.0, sd = 0.0;
    double n_valid = 0.0, n_close_r = 0.0, other_x = 0.0, other_q = 0.0;
    double h1mn = pinf(), h1mx = -pinf(), h2mn = pinf(), h2mx = -pinf();
    for (int s = tid; s < S; s += BLOCK) {
      const double a = (double)wr[(int64_t)s * P.stride_draw];
      const double b = own_lr ? (double)rr[(int64_t)s * P.stride_draw] : a;
      const double x = (double)xr[(int64_t)s * P.stride_draw];
      const double q = x * x;
      const double w = exp(a - mlw);
      sa += w;
      sb = fma(w, x, sb);
      sc = fma(w, q, sc);
      sd = fma(w, w, sd);
      const double rv = exp(b - mlr);  // NaN when the maximum is NaN, or +inf - +inf
      if (rv == rv) {
        n_valid += 1.0;
        if (fabs(rv - 1.0) <= kCloseAtol + kCloseRtol * 1.0) n_close_r += 1.0;
        const double h1 = x * rv, h2 = q * rv;
        h1mn = fmin(h1mn, h1); h1mx = fmax(h1mx, h1);
        h2mn = fmin(h2mn, h2); h2mx = fmax(h2mx, h2);
      }
      if (x != xmn && x != xmx) other_x += 1.0;
      if (q != qmn && q != qmx) other_q += 1.0;
    }
    {
      double a[12] = {sa, sb, sc, sd, n_valid, n_close_r, other_x, other_q, h1mn, h1mx, h2mn, h2mx};
      const int op[12] = {0, 0, 0, 0, 0, 0, 0, 0, 2, 1, 2, 1};
      block_reduce_k<12, BLOCK>(a, op, red);
      sa = a[0]; sb = a[1]; sc = a[2]; sd = a[3]; n_valid = a[4]; n_close_r = a[5]; other_x = a[6]; other_q = a[7];
      h1mn = a[8]; h1mx = a[9]; h2mn = a[10]; h2mx = a[11];
    }
    if (mlr == pinf()) n_close_r = n_valid;  // (a +inf ratio: the valid r are all exp(-inf) = 0, equal to their first)
    // ---- pass 3: how many values of h r lie within the allclose tolerance of each extreme ------------------------
    double c1l = 0.0, c1r = 0.0, c2l = 0.0, c2r = 0.0;
    {
      const double t1l = kCloseAtol + kCloseRtol * fabs(h1mn), t1r = kCloseAtol + kCloseRtol * fabs(h1mx);
      const double t2l = kCloseAtol + kCloseRtol * fabs(h2mn), t2r = kCloseAtol + kCloseRtol * fabs(h2mx);
      for (int s = tid; s < S; s += BLOCK) {
        const double b = (double)rr[(int64_t)s * P.stride_draw];
        const double x = (double)xr[(int64_t)s * P.stride_draw];
        const double rv = exp(b - mlr);
        if (rv == rv) {
          const double h1 = x * rv, h2 = (x * x) * rv;
          if (fabs(h1 - h1mn) <= t1l) c1l += 1.0;
          if (fabs(h1 - h1mx) <= t1r) c1r += 1.0;
          if (fabs(h2 - h2mn) <= t2l) c2l += 1.0;
          if (fabs(h2 - h2mx) <= t2r) c2r += 1.0;
        }
      }
    }
    {
      double a[4] = {c1l, c1r, c2l, c2r};
      const int op[4] = {0, 0, 0, 0};
      block_reduce_k<4, BLOCK>(a, op, red);
      c1l = a[0]; c1r = a[1]; c2l = a[2]; c2r = a[3];
    }
    if (tid == 0)
      e_loo_finish(P, r, n_tail, flags, sa, sb, sc, sd, x0, q0, xdev, qdev, xmn, xmx, qmn, qmx, other_x == 0.0, other_q == 0.0, n_valid,
                   n_close_r, c1l, c1r, c2l, c2r);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Fast path: one WAVEFRONT per observation, ONE pass over the two or three rows (unit draw stride, 16-byte vectors).
//   * the weights are accumulated against a running, wave-uniform maximum of the log-weights (a ballot-guarded branch
//     rescales the four sums when a lane meets a larger one: a handful of times per row once the first 64 vectors are in);
//     the ratios r likewise (OWN: their own maximum and exponential; otherwise r is the weight itself);
//   * what k_hat's guards need from the ORDER of the draws -- are there n values allclose to the extreme? -- is decided
//     from the two most extreme values of each quantity (largest log ratio; smallest and largest of x r and x^2 r), kept
//     per lane with a min/max pair: when the runner-up is outside the tolerance, exactly one value is "close" and no count
//     is needed.  Otherwise (ties, constant weights, a heap of x r below the absolute tolerance) the wave counts, in a
//     second pass over rows that are still in the caches;
//   * "h has exactly two distinct values" (e_loo.py:362-363) is tracked per lane as {lo, hi, saw-a-third} and settled
//     across the lanes against the row's minimum and maximum: exact, no pass;
//   * NaN / +-inf anywhere, or log-weights without a finite maximum: the row goes to e_loo_rows_kernel through the device
//     list (no host round trip), which follows the reference's NaN rules literally.
// exp_tab (|error| <= 2 ulp) stands in for the libm exponential of the general kernel; arguments are clamped at -700, where a
// weight is 1e-304 of the largest one.
// ---------------------------------------------------------------------------------------------------------------------
struct Top2 {  // the two largest values a lane has seen (NEG: the two smallest, stored negated)
  double a, b;
  __device__ __forceinline__ void init() { a = -pinf(); b = -pinf(); }
  __device__ __forceinline__ void put(double v) {
    b = fmax(b, fmin(a, v));
    a = fmax(a, v);
  }
  __device__ __forceinline__ void scale(double f) { a *= f; b *= f; }
  // across the wave: largest and runner-up of the union (ties count: two lanes holding the maximum make them equal)
  __device__ __forceinline__ void merge(int lane, double& m1, double& m2) const {
    m1 = wave_all<R_MAX>(a);
    const unsigned long long who = __ballot(a == m1);
    const int src = __ffsll((long long)who) - 1;
    m2 = wave_all<R_MAX>(lane == src ? b : a);
  }
};

template <typename T, bool OWN>
__global__ __launch_bounds__(256) void e_loo_wave_kernel(ELooParams P) {
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int kW = 4;  // waves per workgroup
  const int tid = threadIdx.x;
  for (int j = tid; j < kTabN; j += kWave * kW) exp_table_entry(tab, j);
  __syncthreads();
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int S = __builtin_amdgcn_readfirstlane(P.n_draws);
  const int n_tail = S < P.tail_len ? S : P.tail_len;
  const int steps = (S + kWave * VEC - 1) / (kWave * VEC);
  const double INF = pinf();
  const auto unpack = [](const v4i& t, double (&o)[VEC]) {
    if constexpr (VEC == 2) {
      o[0] = __hiloint2double(t[1], t[0]);
      o[1] = __hiloint2double(t[3], t[2]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (double)__int_as_float(t[e]);
    }
  };
  const int64_t w0 = (int64_t)blockIdx.x * kW + wv, nw = (int64_t)gridDim.x * kW;
  for (int64_t r = w0; r < P.n_obs; r += nw) {
    const T* xr = reinterpret_cast<const T*>(P.x) + r * P.stride_obs;
    const T* wr = reinterpret_cast<const T*>(P.lw) + r * P.stride_obs;
    const T* rr = reinterpret_cast<const T*>(P.lr) + r * P.stride_obs;
    const int bytes = S * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xr), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wr), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(rr), 0, bytes, 0x00020000);
    const auto load3 = [&](int st, v4i& tx, v4i& tw, v4i& tl) {
      const int off = (st * kWave + lane) * 16;  // (past the end: zeros)
      tx = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      tw = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
      if constexpr (OWN) tl = __builtin_amdgcn_raw_buffer_load_b128(rl, off, 0, 0);
    };
    v4i cx, cw, cl = {0, 0, 0, 0}, nx = {0, 0, 0, 0}, nwv = {0, 0, 0, 0}, nl = {0, 0, 0, 0};
    load3(0, cx, cw, cl);
    if (steps > 1) load3(1, nx, nwv, nl);
    const double x0 = uniform_d((double)xr[0]), q0 = x0 * x0;
    // ---- first vector: the starting maxima (wave-uniform) and the lanes' first values ------------------------------
    double fx[VEC], fw[VEC], fl[VEC];
    unpack(cx, fx);
    unpack(cw, fw);
    if constexpr (OWN) unpack(cl, fl);
    double m, mr;
    {
      double am = -INF, bm = -INF;  // (the first vector is complete in every lane: S >= 64 VEC)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        am = fmax(am, fw[e]);
        if constexpr (OWN) bm = fmax(bm, fl[e]);
      }
      if constexpr (OWN) wave_all2<R_MAX>(am, bm, m, mr);
      else m = mr = wave_all<R_MAX>(am);
    }
    bool bad = !(m > -INF && m < INF) || !(mr > -INF && mr < INF);  // nothing finite to start from (or a NaN / +inf): general kernel
    double sa = 0.0, sb = 0.0, sc = 0.0, sd = 0.0;
    Top2 tb, h1lo, h1hi, h2lo, h2hi;  // largest log ratio; -smallest / largest x r; -smallest / largest x^2 r
    tb.init(); h1lo.init(); h1hi.init(); h2lo.init(); h2hi.init();
    double xlo = fx[0], xhi = fx[0], qlo = fx[0] * fx[0], qhi = qlo;
    bool three_x = false, three_q = false;
    double xdev = 0.0, qdev = 0.0;
    // ALL (every step but the last): the vector lies inside the row in every lane -- none of the selects that turn a draw past
    // the end of the row into one that changes nothing (a fifth of the pass's vector instructions)
    const auto elem = [&](double x, double a, double b, bool valid_in, auto all_c) {
      constexpr bool ALL = decltype(all_c)::value;
      const bool valid = ALL ? true : valid_in;
      if (!valid) {  // past the end of the row: a draw that changes nothing
        x = xlo;
        a = -INF;
        b = -INF;
      }
      const double q = valid ? x * x : qlo;
      bad |= !(q < INF) | (a != a) | (b != b);
      if (__ballot(a > m) != 0ull) {  // a larger log-weight: bring the sums to the new maximum
        const double mn = wave_all<R_MAX>(fmax(a, m));
        const double f = exp_tab(fmax(m - mn, -700.0), tab);
        sa *= f; sb *= f; sc *= f; sd *= f * f;
        if constexpr (!OWN) {
          h1lo.scale(f); h1hi.scale(f); h2lo.scale(f); h2hi.scale(f);
          mr = mn;
        }
        m = mn;
      }
      const double w = valid ? exp_tab(fmax(a - m, -700.0), tab) : 0.0;
      sa += w;
      sb = fma(w, x, sb);
      sc = fma(w, q, sc);
      sd = fma(w, w, sd);
      double rv = w;
      if constexpr (OWN) {
        if (__ballot(b > mr) != 0ull) {
          const double mn = wave_all<R_MAX>(fmax(b, mr));
          const double f = exp_tab(fmax(mr - mn, -700.0), tab);
          h1lo.scale(f); h1hi.scale(f); h2lo.scale(f); h2hi.scale(f);
          mr = mn;
        }
        rv = exp_tab(fmax(b - mr, -700.0), tab);
      }
      const double lb = OWN ? b : a;
      tb.put(valid ? lb : -INF);
      const double h1 = x * rv, h2 = q * rv;
      h1lo.put(valid ? -h1 : -INF);
      h1hi.put(valid ? h1 : -INF);
      h2lo.put(valid ? -h2 : -INF);
      h2hi.put(valid ? h2 : -INF);
      // distinct values of x and x^2 in this lane: a third one shows as a value strictly inside [lo, hi], or outside it
      // when lo and hi already differ
      {
        const bool out = (x < xlo) | (x > xhi), in = (x != xlo) & (x != xhi);
        three_x |= in & (!out | (xlo != xhi));
        xlo = fmin(xlo, x);
        xhi = fmax(xhi, x);
        const bool outq = (q < qlo) | (q > qhi), inq = (q != qlo) & (q != qhi);
        three_q |= inq & (!outq | (qlo != qhi));
        qlo = fmin(qlo, q);
        qhi = fmax(qhi, q);
      }
      xdev = fmax(xdev, fabs(x - x0));
      qdev = fmax(qdev, fabs(q - q0));
    };
#pragma unroll 1
    for (int st = 0; st < steps - 1; ++st) {
      double vx[VEC], vw[VEC], vl[VEC];
      unpack(cx, vx);
      unpack(cw, vw);
      if constexpr (OWN) unpack(cl, vl);
      cx = nx; cw = nwv; cl = nl;
      if (st + 2 < steps) load3(st + 2, nx, nwv, nl);
#pragma unroll
      for (int e = 0; e < VEC; ++e) elem(vx[e], vw[e], OWN ? vl[e] : vw[e], true, std::true_type{});
    }
    {  // the last step: some lanes' vectors lie past the end of the row
      const int st = steps - 1;
      double vx[VEC], vw[VEC], vl[VEC];
      unpack(cx, vx);
      unpack(cw, vw);
      if constexpr (OWN) unpack(cl, vl);
      const bool valid = (st * kWave + lane) * VEC < S;  // (S is a multiple of VEC: a vector is inside the row or past it)
#pragma unroll
      for (int e = 0; e < VEC; ++e) elem(vx[e], vw[e], OWN ? vl[e] : vw[e], valid, std::false_type{});
    }
    // ---- across the lanes ----------------------------------------------------------------------------------------
    double SA, SB, SC, SD;
    wave_all4<R_SUM>(sa, sb, sc, sd, SA, SB, SC, SD);
    double xmn, xmx, qmn, qmx, nxmn, nqmn;
    wave_all4<R_MAX>(-xlo, xhi, -qlo, qhi, nxmn, xmx, nqmn, qmx);
    xmn = -nxmn; qmn = -nqmn;
    double XD, QD;
    wave_all2<R_MAX>(xdev, qdev, XD, QD);
    const bool two_x = __ballot(three_x | ((xlo != xmn) & (xlo != xmx)) | ((xhi != xmn) & (xhi != xmx))) == 0ull;
    const bool two_q = __ballot(three_q | ((qlo != qmn) & (qlo != qmx)) | ((qhi != qmn) & (qhi != qmx))) == 0ull;
    double b1, b2, l1a, l1b, r1a, r1b, l2a, l2b, r2a, r2b;
    tb.merge(lane, b1, b2);
    h1lo.merge(lane, l1a, l1b);
    h1hi.merge(lane, r1a, r1b);
    h2lo.merge(lane, l2a, l2b);
    h2hi.merge(lane, r2a, r2b);
    const double h1mn = -l1a, h1mx = r1a, h2mn = -l2a, h2mx = r2a;
    bad |= !(m < INF) | !(mr < INF);
    const bool defer = __ballot(bad) != 0ull;
    if (defer) {
      if (lane == 0) {
        const unsigned long long idx = atomicAdd(P.slow_count, 1ull);
        P.slow_list[idx] = (unsigned)r;
      }
      continue;
    }
    // ---- is any of the five counts needed?  (1 % of slack: a runner-up right at the tolerance is counted, not guessed) ----
    const auto near = [](double runner, double ext) { return fabs(runner - ext) <= 1.01 * (kCloseAtol + kCloseRtol * fabs(ext)); };
    const double rv2 = exp(b2 - b1);  // the runner-up among the ratios (the largest is 1)
    double n_close_r = 1.0, c1l = 1.0, c1r = 1.0, c2l = 1.0, c2r = 1.0;
    if (near(rv2, 1.0) || near(-l1b, h1mn) || near(r1b, h1mx) || near(-l2b, h2mn) || near(r2b, h2mx)) {
      double cr = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
      const double t1l = kCloseAtol + kCloseRtol * fabs(h1mn), t1r = kCloseAtol + kCloseRtol * fabs(h1mx);
      const double t2l = kCloseAtol + kCloseRtol * fabs(h2mn), t2r = kCloseAtol + kCloseRtol * fabs(h2mx);
#pragma unroll 1
      for (int st = 0; st < steps; ++st) {
        v4i tx, tw, tl = {0, 0, 0, 0};
        load3(st, tx, tw, tl);
        double vx[VEC], vl[VEC];
        unpack(tx, vx);
        unpack(OWN ? tl : tw, vl);
        if ((st * kWave + lane) * VEC < S) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const double rv = exp_tab(fmax(vl[e] - mr, -700.0), tab);
            const double h1 = vx[e] * rv, h2 = (vx[e] * vx[e]) * rv;
            cr += (fabs(rv - 1.0) <= kCloseAtol + kCloseRtol) ? 1.0 : 0.0;
            a1 += (fabs(h1 - h1mn) <= t1l) ? 1.0 : 0.0;
            a2 += (fabs(h1 - h1mx) <= t1r) ? 1.0 : 0.0;
            a3 += (fabs(h2 - h2mn) <= t2l) ? 1.0 : 0.0;
            a4 += (fabs(h2 - h2mx) <= t2r) ? 1.0 : 0.0;
          }
        }
      }
      wave_all4<R_SUM>(a1, a2, a3, a4, c1l, c1r, c2l, c2r);
      n_close_r = wave_all<R_SUM>(cr);
    }
    if (lane == 0) {
      // (what the end of a row needs of the launch's arguments -- five output pointers, the tail length and the constants derived
      // from it -- is read from the argument block HERE, behind an opaque pointer: hoisted above the row loop it sat in scalar
      // registers the loop does not have, 19 of them spilled into vector lanes -- the form that came back wrong in round 4's fit
      // kernel; tests/test_kernel_resources.py holds this kernel to none)
#if defined(__HIP_DEVICE_COMPILE__)
      typedef const __attribute__((address_space(4))) ELooParams* ArgPtr;
      ArgPtr qp = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(qp));
      const ELooParams Pl = *qp;
#else
      const ELooParams& Pl = P;
#endif
      const int nt = S < Pl.tail_len ? S : Pl.tail_len;
      e_loo_finish(Pl, r, nt, 0u, SA, SB, SC, SD, x0, q0, XD, QD, xmn, xmx, qmn, qmx, two_x, two_q, (double)S, n_close_r, c1l, c1r,
                   c2l, c2r);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Weighted quantiles (e_loo.py:468-515 -> `_weighted_quantile`, 534-554), one 256-thread workgroup per observation.
//
// The reference argsorts the draws, accumulates the sorted weights and interpolates between the two draws that bracket
// `prob`.  No sort is needed for that: what it reads off the sorted arrays is (a) the smallest draw v whose cumulative
// weight reaches prob, (b) the weight strictly below v and at v, (c) the largest draw below v.  (a) is an MSB-first radix
// descent on order-preserving 64-bit keys, 8 bits per pass, with a 256-bin LDS histogram OF WEIGHTS (ds_add_f64) -- the
// weighted twin of the general kernel's selection (pla_rows.h) -- which also yields (b); (c) is one more pass.  Constant
// weights take np.quantile's branch (e_loo.py:536-537): the same descent on counts, numpy's `linear` interpolation.
// Equal draws: the descent finds the group as a whole; inside it the reference's element-by-element walk is followed -- only
// the group's FIRST member (here: lowest draw index; the reference: wherever its unstable argsort puts it) interpolates from
// the value below, a target crossed at a later member returns the value itself.
// ---------------------------------------------------------------------------------------------------------------------
struct EQuantParams {
  const void* x;
  const void* lw;
  int64_t n_obs;
  int n_draws;
  int64_t stride_obs, stride_draw;
  const double* probs;  // [n_probs] device
  int n_probs;
  double* out;          // [n_obs][n_probs]
  // rows the wave-per-observation kernel does not take (non-finite or constant draws, too many draws in a level's histogram
  // bins) are listed here and redone by the 512-thread kernel, which walks the list when one is given
  unsigned* slow_list = nullptr;
  unsigned long long* slow_count = nullptr;
};

// smallest key whose cumulative mass (sum of the masses of the draws with key <= it) reaches `target`; *below = mass strictly
// below that key, *at = mass at it.  `each(f)` calls f(key, mass) for every draw of this thread.  Returns false when the
// total never reaches the target.
// After two passes (16 bits of the key: sign, exponent, four bits of mantissa) the bin that holds the target rarely has more
// than a few dozen draws: they are collected into an LDS list and settled there (every thread of the list takes one element
// and adds up the masses at or below it) instead of six more passes over all the draws.  A list that does not fit (more than
// 256 draws share the prefix) goes on with the radix descent and tries again one byte further down.
struct QuantList {
  uint64_t key[256];
  double mass[256];
  unsigned long long best;
  int count;
};
template <int BLOCK, class Each>
__device__ __forceinline__ bool mass_select(Each each, const double target, double* hist, double* red, QuantList* ql, uint64_t* key_out,
                                            double* below, double* at) {
  static_assert(BLOCK >= 256, "one thread per list entry");
  const int tid = threadIdx.x;
  uint64_t prefix = 0;
  double base = 0.0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0.0;
    __syncthreads();
    each([&](const uint64_t k, const double mass) {
      if (shift == 56 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], mass);
    });
    __syncthreads();
    // the first wave walks the 256 bins, four per lane: in-lane running sums on top of a shuffle scan of the lane totals (a
    // fixed order of additions, so the result is reproducible; one thread reading bin after bin was 256 dependent LDS round
    // trips per level -- 30 000 cycles -- and most of this kernel's time)
    if (tid < kWave) {
      const double4 h = *reinterpret_cast<const double4*>(&hist[4 * tid]);
      const double local = ((h.x + h.y) + h.z) + h.w;
      double incl = local;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const double up = __shfl_up(incl, o, kWave);
        if (tid >= o) incl += up;
      }
      double before = __shfl_up(incl, 1, kWave);  // mass in the bins of the lanes below
      before = base + (tid == 0 ? 0.0 : before);
      const double c1 = before + h.x, c2 = c1 + h.y, c3 = c2 + h.z, c4 = c3 + h.w;
      const int i = c1 >= target ? 0 : (c2 >= target ? 1 : (c3 >= target ? 2 : (c4 >= target ? 3 : 4)));
      const unsigned long long hit = __ballot(i < 4);
      const int first = hit ? __ffsll((long long)hit) - 1 : kWave - 1;
      if (tid == first) {
        if (hit) {
          red[0] = i == 0 ? before : (i == 1 ? c1 : (i == 2 ? c2 : c3));
          red[1] = (double)(4 * tid + i);
          red[2] = i == 0 ? h.x : (i == 1 ? h.y : (i == 2 ? h.z : h.w));
        } else {
          red[0] = c4;  // (the total never reaches the target)
          red[1] = 256.0;
          red[2] = 0.0;
        }
      }
    }
    __syncthreads();
    const int d = (int)red[1];
    base = red[0];
    const double here = red[2];
    __syncthreads();
    if (d >= 256) return false;
    prefix = (prefix << 8) | (uint64_t)d;
    if (shift <= 48 && shift > 0) {
      if (tid == 0) {
        ql->count = 0;
        ql->best = ~0ull;
      }
      __syncthreads();
      each([&](const uint64_t k, const double mass) {
        if ((k >> shift) == prefix) {
          const int idx = atomicAdd(&ql->count, 1);
          if (idx < 256) {
            ql->key[idx] = k;
            ql->mass[idx] = mass;
          }
        }
      });
      __syncthreads();
      const int n = ql->count;
      if (n <= 256) {  // (wave-uniform: every thread reads the same count)
        uint64_t mine = ~0ull;
        double upto = base, under = base;
        if (tid < n) {
          mine = ql->key[tid];
          for (int j = 0; j < n; ++j) {
            const uint64_t kj = ql->key[j];
            const double mj = ql->mass[j];
            upto += (kj <= mine) ? mj : 0.0;
            under += (kj < mine) ? mj : 0.0;
          }
        }
        const bool reaches = tid < n && upto >= target;
        if (reaches) atomicMin(&ql->best, (unsigned long long)mine);
        __syncthreads();
        const uint64_t best = (uint64_t)ql->best;
        if (best == ~0ull) return false;  // (the total never reaches the target)
        if (reaches && mine == best) {     // (equal draws compute the same two numbers)
          red[0] = under;
          red[2] = upto - under;
        }
        __syncthreads();
        *key_out = best;
        *below = red[0];
        *at = red[2];
        __syncthreads();
        return true;
      }
    }
    if (shift == 0) {
      *key_out = prefix;
      *below = base;
      *at = here;
    }
  }
  return true;
}

// The same selection from ONE histogram of the row shared by every level asked for: kQBins bins, linear in x between the
// row's extremes, filled once (weights, or counts for np.quantile's branch) and turned into running sums; a level then costs a
// parallel look-up of its bin, one trip over the draws held in registers to collect that bin's members (a handful) and the
// settling of that list -- instead of two or three radix passes of one LDS atomic per draw EACH (4000 atomics a pass).
// Returns 1: found; 0: the total never reaches the target; -1: more than 256 draws share the bin (very uneven rows: the caller
// falls back to the radix descent above).
// NS sums and NM maxima over the workgroup in ONE LDS exchange (two barriers whatever NS + NM): what this kernel costs are its
// block-wide reductions (~1.5 us each), a dozen per row and level when every value goes round by itself.  `lds`: (NS + NM)
// doubles per wave.  Maxima ignore NaN (callers flag NaN separately); a minimum goes in negated.
template <int NS, int NM, int BLOCK>
__device__ __forceinline__ void block_reduce_mix(double (&sv)[NS > 0 ? NS : 1], double (&mv)[NM > 0 ? NM : 1], double* lds) {
  constexpr int NW = BLOCK / kWave, N = NS + NM;
  const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double r = wave_all<R_SUM>(sv[i]);
    if (lane == 0) lds[w * N + i] = r;
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    const double r = wave_all<R_MAX>(mv[i]);
    if (lane == 0) lds[w * N + NS + i] = r;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) a += lds[k * N + i];
    sv[i] = a;
  }
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    double a = lds[NS + i];
#pragma unroll
    for (int k = 1; k < NW; ++k) a = fmax(a, lds[k * N + NS + i]);
    mv[i] = a;
  }
  __syncthreads();
}

constexpr int kQBins = 2048;
__device__ __forceinline__ int qbin_of(double x, double x0, double scale) {
  const int b = (int)((x - x0) * scale);
  return b < 0 ? 0 : (b > kQBins - 1 ? kQBins - 1 : b);
}
// Scratch of one hist_select call.  Two of them take turns: a call leaves its results in its own and resets the OTHER one for the
// call after it, so that no barrier is spent on "everybody has read this before it is written again" -- four per call, not seven.
struct HistSelectScratch {
  double red[4];  // [0] mass below the bin / below the draw, [1] the bin (-1: none), [2] mass at the draw
  QuantList ql;
};
__device__ __forceinline__ void hist_select_reset(HistSelectScratch* h) {
  h->red[1] = -1.0;
  h->ql.count = 0;
  h->ql.best = ~0ull;
}
// `mine` must have been reset (by the call before, or by the caller ahead of a barrier); `other` is reset here for the next call.
template <int BLOCK, class Each>
__device__ __forceinline__ int hist_select(Each each, const double target, const double* cum, const double x0, const double scale,
                                           HistSelectScratch* mine, HistSelectScratch* other, uint64_t* key_out, double* below, double* at) {
  const int tid = threadIdx.x;
  constexpr int PER = kQBins / BLOCK;
  double* const red = mine->red;
  QuantList* const ql = &mine->ql;
  // (the rare ways out: the other scratch is reset behind a barrier of its own)
  const auto leave = [&](const int code) {
    if (tid == 0) hist_select_reset(other);
    __syncthreads();
    return code;
  };
  {  // the bin in which the running sum first reaches the target (exactly one thread finds it)
    double prev = tid == 0 ? 0.0 : cum[tid * PER - 1];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const double c = cum[tid * PER + i];
      if (prev < target && c >= target) {
        red[0] = prev;
        red[1] = (double)(tid * PER + i);
      }
      prev = c;
    }
  }
  __syncthreads();
  const int tb = (int)red[1];
  const double base = red[0];
  if (tb < 0) return leave(0);
  each([&](const uint64_t k, const double mass) {
    if (qbin_of(val_of(k), x0, scale) == tb) {
      const int idx = atomicAdd(&ql->count, 1);
      if (idx < 256) {
        ql->key[idx] = k;
        ql->mass[idx] = mass;
      }
    }
  });
  __syncthreads();
  const int n = ql->count;
  if (n > 256) return leave(-1);
  uint64_t mykey = ~0ull;
  double upto = base, under = base;
  if (tid < n) {
    mykey = ql->key[tid];
    for (int j = 0; j < n; ++j) {
      const uint64_t kj = ql->key[j];
      const double mj = ql->mass[j];
      upto += (kj <= mykey) ? mj : 0.0;
      under += (kj < mykey) ? mj : 0.0;
    }
  }
  const bool reaches = tid < n && upto >= target;
  if (reaches) atomicMin(&ql->best, (unsigned long long)mykey);
  __syncthreads();
  const uint64_t best = (uint64_t)ql->best;
  if (best == ~0ull) return leave(-1);  // (rounding at the bin's edge: let the radix descent decide)
  if (reaches && mykey == best) {  // (equal draws compute the same two numbers)
    red[0] = under;
    red[2] = upto - under;
  }
  if (tid == 0) hist_select_reset(other);
  __syncthreads();
  *key_out = best;
  *below = red[0];
  *at = red[2];
  return 1;
}

// Over all rows, or -- behind the wave-per-observation kernel below -- over the rows that kernel listed (P.slow_list).
// (Rounds 2-3 ran a second, histogram-only instantiation of this kernel in front of it, compiled for 128 registers with 132 bytes
// of scratch; the wave kernel took its place in round 4.)
template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK, 1) void e_loo_quantile_kernel(EQuantParams P) {
  __shared__ __attribute__((aligned(32))) double cum2k[kQBins];
  __shared__ __attribute__((aligned(32))) double scan[BLOCK];
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];  // table-driven exponential (pla_math.h): a third of libm's registers
  for (int j = threadIdx.x; j < kTabN; j += BLOCK) exp_table_entry(tab, j);
  __syncthreads();
  // w = e^(lw - max lw); a NaN weight or a non-finite maximum makes every weight NaN in the reference (the sum is poisoned below)
  const auto wexp = [&](double d) { return exp_tab(fmax(d, -700.0), tab); };
  __shared__ __attribute__((aligned(32))) double hist[256];
  __shared__ double red[BLOCK / kWave > 4 ? BLOCK / kWave : 4];
  __shared__ double mix[4 * (BLOCK / kWave)];
  __shared__ QuantList qlist;             // (the radix descent's)
  __shared__ HistSelectScratch hsel[2];   // (hist_select's, in turns)
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  const int64_t n_rows = P.slow_list ? (int64_t)*P.slow_count : P.n_obs;
  for (int64_t ri = blockIdx.x; ri < n_rows; ri += gridDim.x) {
    const int64_t r = P.slow_list ? (int64_t)P.slow_list[ri] : ri;
    const T* xr = reinterpret_cast<const T*>(P.x) + r * P.stride_obs;
    const T* wr = reinterpret_cast<const T*>(P.lw) + r * P.stride_obs;
    const auto xat = [&](int s) { return (double)xr[(int64_t)s * P.stride_draw]; };
    const auto key_at = [&](int s) { return key_of(xat(s)); };
    // normalised weights w = exp(lw - logsumexp(lw)) (e_loo.py:557-559, 473): maximum, sum, and whether they are all close.
    // Each thread's draws -- order-preserving key and normalised weight -- stay in its registers for everything that follows
    // (up to kKeep per thread: S <= 4096 at 256 threads; longer rows re-read and re-evaluate): ONE trip over the two rows with
    // all its loads in flight together and one exponential per draw, then LDS atomics only -- instead of four trips with
    // dependent loads before the first radix pass and an exponential and a division per draw in every pass after it.
    constexpr int kKeep = 4096 / BLOCK;  // (512 threads: 8 draws each -- the kernel then fits four waves per SIMD)
    const bool kept = S <= kKeep * BLOCK;
    uint64_t kreg[kKeep];
    double wreg[kKeep];
    double mlw = -pinf(), xmax = -pinf(), xmin = pinf(), sa = 0.0, dev = 0.0, w0, wtot_kept = 0.0;
    unsigned nanw = 0;
    if (kept) {
      double xv[kKeep];
#pragma unroll
      for (int j = 0; j < kKeep; ++j) {  // (all 32 loads of the thread go out before anything waits)
        const int s = tid + j * BLOCK;
        const int sc = s < S ? s : 0;
        xv[j] = (double)xr[(int64_t)sc * P.stride_draw];
        wreg[j] = (double)wr[(int64_t)sc * P.stride_draw];
      }
#pragma unroll
      for (int j = 0; j < kKeep; ++j) {
        const bool in = tid + j * BLOCK < S;
        if (in && wreg[j] != wreg[j]) nanw = 1u;
        mlw = in ? fmax(mlw, wreg[j]) : mlw;
        xmax = in ? fmax(xmax, xv[j]) : xmax;
        xmin = in ? fmin(xmin, xv[j]) : xmin;
        kreg[j] = in ? key_of(xv[j]) : ~0ull;
      }
      {
        double none[1] = {0.0}, mx4[4] = {mlw, xmax, -xmin, nanw ? 1.0 : 0.0};
        block_reduce_mix<0, 4, BLOCK>(none, mx4, mix);
        mlw = mx4[0]; xmax = mx4[1]; xmin = -mx4[2]; nanw = mx4[3] > 0.0 ? 1u : 0u;
      }
#pragma unroll
      for (int j = 0; j < kKeep; ++j) {
        wreg[j] = tid + j * BLOCK < S ? wexp(wreg[j] - mlw) : 0.0;
        sa += wreg[j];
      }
      sa = block_reduce<OpSum, BLOCK>(sa, red);
      if (nanw || !(fabs(mlw) < pinf())) sa = qnan();
      const double inv_sa = 1.0 / sa;
      w0 = wexp((double)wr[0] - mlw) * inv_sa;
      double wsum = 0.0;
#pragma unroll
      for (int j = 0; j < kKeep; ++j) {
        wreg[j] = wreg[j] * inv_sa;
        dev = tid + j * BLOCK < S ? fmax(dev, fabs(wreg[j] - w0)) : dev;
        wsum += tid + j * BLOCK < S ? wreg[j] : 0.0;
      }
      {  // the largest deviation from the first weight and the sum of the normalised weights (e_loo.py:542) in one exchange
        double s1[1] = {wsum}, m1[1] = {dev};
        block_reduce_mix<1, 1, BLOCK>(s1, m1, mix);
        wtot_kept = s1[0];
        dev = m1[0];
      }
    } else {
      for (int s = tid; s < S; s += BLOCK) {
        const double a = (double)wr[(int64_t)s * P.stride_draw];
        if (a != a) nanw = 1u;
        mlw = fmax(mlw, a);
        const double x = xat(s);
        xmax = fmax(xmax, x);
        xmin = fmin(xmin, x);
      }
      mlw = block_reduce<OpMax, BLOCK>(mlw, red);
      xmax = block_reduce<OpMax, BLOCK>(xmax, red);
      xmin = block_reduce<OpMin, BLOCK>(xmin, red);
      nanw = block_or_bits<BLOCK>(nanw, red);
      for (int s = tid; s < S; s += BLOCK) sa += wexp((double)wr[(int64_t)s * P.stride_draw] - mlw);
      sa = block_reduce<OpSum, BLOCK>(sa, red);
      if (nanw || !(fabs(mlw) < pinf())) sa = qnan();
      w0 = wexp((double)wr[0] - mlw) / sa;
      for (int s = tid; s < S; s += BLOCK) dev = fmax(dev, fabs(wexp((double)wr[(int64_t)s * P.stride_draw] - mlw) / sa - w0));
      dev = block_reduce<OpMax, BLOCK>(dev, red);
    }
    const bool flat = !nanw && dev <= kCloseAtol + kCloseRtol * fabs(w0);                 // e_loo.py:536
    const auto wat = [&](int s) { return wexp((double)wr[(int64_t)s * P.stride_draw] - mlw) / sa; };
    const auto each = [&](auto f) {  // f(key, weight) for every draw of this thread
      if (kept) {
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
          if (tid + j * BLOCK < S) f(kreg[j], wreg[j]);
      } else {
        for (int s = tid; s < S; s += BLOCK) f(key_at(s), wat(s));
      }
    };
    const auto each_count = [&](auto f) { each([&](const uint64_t k, const double) { f(k, 1.0); }); };
    const auto each_s = [&](auto f) {  // f(key, weight, draw index)
      if (kept) {
#pragma unroll
        for (int j = 0; j < kKeep; ++j)
          if (tid + j * BLOCK < S) f(kreg[j], wreg[j], tid + j * BLOCK);
      } else {
        for (int s = tid; s < S; s += BLOCK) f(key_at(s), wat(s), s);
      }
    };
    double wtot = wtot_kept;
    if (!flat && !kept) {
      each([&](const uint64_t, const double w) { wtot += w; });
      wtot = block_reduce<OpSum, BLOCK>(wtot, red);                                       // e_loo.py:542: cumsum / sum
    }
    // one histogram of the row for every level (hist_select): rows kept in registers with finite, distinct extremes
    const bool hist_ok = kept && xmax > xmin && xmax - xmin < 1e300 && xmin > -1e300;
    const double qscale = hist_ok ? (double)kQBins / (xmax - xmin) : 0.0;
    if (hist_ok) {
      for (int i = tid; i < kQBins; i += BLOCK) cum2k[i] = 0.0;
      __syncthreads();
      each([&](const uint64_t k, const double w) { atomicAdd(&cum2k[qbin_of(val_of(k), xmin, qscale)], flat ? 1.0 : w); });
      __syncthreads();
      // running sums in place: each thread its kQBins / BLOCK bins, a block scan of the thread totals in between
      constexpr int PER = kQBins / BLOCK;
      double loc[PER], run = 0.0;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        run += cum2k[tid * PER + i];
        loc[i] = run;
      }
      scan[tid] = run;
      __syncthreads();
      if (tid < kWave) {  // exclusive scan of the BLOCK thread totals by the first wave, BLOCK / 64 per lane
        constexpr int Q = BLOCK / kWave;
        double part[Q], tot = 0.0;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          part[i] = tot;
          tot += scan[Q * tid + i];
        }
        double incl = tot;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const double up = __shfl_up(incl, o, kWave);
          if (tid >= o) incl += up;
        }
        const double before = incl - tot;
#pragma unroll
        for (int i = 0; i < Q; ++i) scan[Q * tid + i] = before + part[i];
      }
      __syncthreads();
      const double off = scan[tid];
#pragma unroll
      for (int i = 0; i < PER; ++i) cum2k[tid * PER + i] = off + loc[i];
      if (tid == 0) hist_select_reset(&hsel[0]);
      __syncthreads();
    }
    int turn = 0;  // whose turn it is among hist_select's two scratches
    for (int ip = 0; ip < P.n_probs; ++ip) {
      const double prob = P.probs[ip];
      double res;
      uint64_t kv = 0;
      double below = 0.0, at = 0.0;
      if (flat) {
        // np.quantile(x, prob), method "linear": virtual index (S - 1) prob between the order statistics lo and lo + 1
        const double virt = (double)(S - 1) * prob;
        const double lo = floor(virt), t = virt - lo;
        int hs = hist_ok ? hist_select<BLOCK>(each_count, lo + 1.0, cum2k, xmin, qscale, &hsel[turn], &hsel[turn ^ 1], &kv, &below, &at) : -1;
        turn ^= hist_ok ? 1 : 0;
        if (hs < 0) mass_select<BLOCK>(each_count, lo + 1.0, hist, red, &qlist, &kv, &below, &at);
        const double a = val_of(kv);
        double b = a;
        if (below + at < lo + 2.0 && lo + 1.0 < (double)S) {  // the next order statistic is the next distinct value
          double nxt = pinf();
          each([&](const uint64_t k, const double) {
            if (k > kv) nxt = fmin(nxt, val_of(k));
          });
          b = block_reduce<OpMin, BLOCK>(nxt, red);
        }
        const double diff = b - a;
        res = (t >= 0.5) ? b - diff * (1.0 - t) : a + diff * t;                           // numpy's _lerp
        if (t == 0.0) res = a;
      } else {
        const int hs = hist_ok ? hist_select<BLOCK>(each, prob * wtot, cum2k, xmin, qscale, &hsel[turn], &hsel[turn ^ 1], &kv, &below, &at) : -1;
        turn ^= hist_ok ? 1 : 0;
        bool found = hs == 1;
        if (hs < 0) found = mass_select<BLOCK>(each, prob * wtot, hist, red, &qlist, &kv, &below, &at);
        if (!found) {
          res = xmax;                                                                     // 545-546
        } else {
          const double v = val_of(kv);
          double prev = -pinf();
          double sfirst = 1e300;
          each_s([&](const uint64_t k, const double, const int si) {
            if (k < kv) prev = fmax(prev, val_of(k));
            if (k == kv) sfirst = fmin(sfirst, (double)si);
          });
          {  // the largest draw below v and the first (lowest index) of the draws equal to v, in one exchange
            double none[1] = {0.0}, m2[2] = {prev, -sfirst};
            block_reduce_mix<0, 2, BLOCK>(none, m2, mix);
            prev = m2[0];
            sfirst = -m2[1];
          }
          // Equal draws: the reference walks the SORTED draws one by one (542-554), so inside a group of equal draws only
          // its first member interpolates from the value below -- with its own weight -- and a target crossed at any later
          // member has x1 == x_sorted[wi] and returns v exactly.  (Collapsing the group into one draw of the combined weight
          // gave 4.36 where the reference gives 5.0 on count data.)  The reference's order inside the group is that of an
          // unstable argsort; here it is draw order: the first member is the tied draw with the lowest index.
          // (its weight: the one thread that holds that draw says so -- one barrier; mix[] is free behind block_reduce_mix)
          each_s([&](const uint64_t, const double w, const int si) {
            if ((double)si == sfirst) mix[0] = w;
          });
          __syncthreads();
          const double wfirst = mix[0];
          if (below == 0.0 && prev == -pinf()) res = v;                                   // wi == 0: 548-550
          else if (!(below + wfirst >= prob * wtot)) res = v;                             // crossed inside the group of equal draws
          else {
            const double w1 = below / wtot, wwi = (below + wfirst) / wtot;                // 552
            res = prev + (v - prev) * (prob - w1) / (wwi - w1);                           // 554
          }
          (void)at;
        }
      }
      if (tid == 0) P.out[r * P.n_probs + ip] = res;
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weighted quantiles, fast path (round 4): ONE WAVEFRONT per observation, ONE trip over the two rows, no workgroup barrier
// anywhere in the row loop.
//
// The 512-thread kernel above spends its time in ~30 block-wide exchanges per row (11.1 ms per 200 k x 4000
// x 3 levels, 0.14 of what the two matrices cost to read).  Here a wave owns the row:
//   * the draws and the log-weights pass through the registers once, in batches of eight 16-byte vectors each; the weights are
//     taken against a RUNNING maximum of the log-weights that is brought up to date once per batch (wave-uniform; when it moves,
//     the histogram so far is rescaled: twice per row on average);
//   * ONE histogram of the row serves every level: 1024 bins linear in x between two bounds taken from the row's first 512
//     draws and widened (draws outside fall into the end bins, which are settled like any other), the weight AND the number of the
//     draws per bin (ds_add_f64 / ds_add_u32), both turned into running sums by a 16-bins-per-lane scan; every draw's bin number
//     stays in LDS, two bytes each, in the order the lane met them;
//   * a level = two look-ups in those sums -- the bin tb in which the cumulative weight reaches prob * total, and the last
//     non-empty bin below it (by COUNT: a draw of negligible weight still is the reference's x_sorted[wi - 1]) -- one trip over
//     the lane's bin numbers that collects the members of those two bins (a scalar branch skips the slots without one: most),
//     their draws and log-weights read again from the matrices, and the settling of that list across the lanes: smallest member
//     whose cumulative weight reaches the target, the weight below it, the first (lowest draw index) of the draws equal to it,
//     the largest member below it -- exactly what the reference reads off its sorted arrays (e_loo.py:541-554).
// Rows it does not take go on the device list for the general kernel above: non-finite draws or log-weights, constant
// draws, weights that are all close (np.quantile's branch, e_loo.py:536-537), more than 64 draws in the two bins of a level, and
// the measure-zero cases in which rounding puts the crossing on the other side of a bin edge.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kQWBins = 1024;
constexpr int kQWBatch = 8;                          // vectors of each matrix in flight per lane
struct QuantWaveSmem {
  double hw[kQWBins];                                // weight per bin -> running sums
  unsigned short hc[kQWBins];                        // draws per bin -> running sums (a row has at most 4096 draws; filled by 32-bit atomics on bin pairs)
  unsigned short bins[kWave * (kWaveSlots + 4)];     // bin of every draw: lane l's slots at l * 68 (136 bytes: 8-byte aligned rows,
                                                     // four lanes to a bank when all write the same slot)
  int ms[kWave];                                     // members of the two bins of a level: draw index
};

template <typename T>
__global__ __launch_bounds__(256, 2) void e_loo_quantile_wave_kernel(EQuantParams P) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int NQ = kWaveSlots / VEC;
  constexpr int kW = 4;
  constexpr int kRow = kWaveSlots + 4;  // bin numbers per lane in LDS (padded)
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  __shared__ __attribute__((aligned(16))) QuantWaveSmem smem[kW];
  const int tid = threadIdx.x;
  for (int j = tid; j < kTabN; j += kWave * kW) exp_table_entry(tab, j);
  __syncthreads();
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  QuantWaveSmem& sm = smem[wv];
  const int S = __builtin_amdgcn_readfirstlane(P.n_draws);
  const int n_probs = __builtin_amdgcn_readfirstlane(P.n_probs);
  const int nvec = S / VEC, qfull = nvec / kWave, qrem = nvec - qfull * kWave;
  const int nval = qfull + (lane < qrem ? 1 : 0);  // vectors of the row in this lane: vector q is inside the row iff q < nval
  const double INF = pinf();
  const auto wexp = [&](double d) { return exp_tab(fmax(d, -700.0), tab); };
  const auto unpack = [](const v4i& t, double (&o)[VEC]) {
    if constexpr (VEC == 2) {
      o[0] = __hiloint2double(t[1], t[0]);
      o[1] = __hiloint2double(t[3], t[2]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (double)__int_as_float(t[e]);
    }
  };
  constexpr int PER = kQWBins / kWave;  // 16 bins per lane in the scans
  const int64_t w0 = (int64_t)blockIdx.x * kW + wv, nw = (int64_t)gridDim.x * kW;
#pragma unroll 1
  for (int64_t r = w0; r < P.n_obs; r += nw) {
    // (the arguments are read from the kernel's argument block where they are used -- scalar loads from constant memory -- not
    // held in scalar registers across the row loop: the loop has none to spare, and scalars spilled into vector lanes have come
    // back wrong in this build's fit kernel; tests/test_kernel_resources.py holds this kernel to zero such spills)
    typedef const __attribute__((address_space(4))) EQuantParams* ArgPtr;
    ArgPtr qp = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(qp));
    const auto decline = [&]() {
      if (lane == 0) qp->slow_list[atomicAdd(qp->slow_count, 1ull)] = (unsigned)r;
    };
    const T* xr = reinterpret_cast<const T*>(qp->x) + r * qp->stride_obs;
    const T* wr = reinterpret_cast<const T*>(qp->lw) + r * qp->stride_obs;
    const int bytes = S * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xr), 0, bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wr), 0, bytes, 0x00020000);
    v4i tx[kQWBatch], tw[kQWBatch];
    const auto load_batch = [&](int q0) {
#pragma unroll
      for (int u = 0; u < kQWBatch; ++u) {
        tx[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16, (q0 + u) * (kWave * 16), 2);  // (read once: non-temporal)
        tw[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (q0 + u) * (kWave * 16), 2);
      }
    };
    load_batch(0);
    wave_sync();  // (the previous row is done with the scratch)
#pragma unroll 1
    for (int b = lane; b < kQWBins; b += kWave) {
      sm.hw[b] = 0.0;
      sm.hc[b] = 0;
    }
    // ---- bounds of the bins from the first batch (8 x 64 vectors: the row's first 512 VEC draws, all inside the row) ---------
    double blo, bhi;
    {
      double lo = INF, hi = -INF;
#pragma unroll
      for (int u = 0; u < kQWBatch; ++u) {
        double x[VEC];
        unpack(tx[u], x);
        const bool valid = u < nval;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          lo = valid ? fmin(lo, x[e]) : lo;
          hi = valid ? fmax(hi, x[e]) : hi;
        }
      }
      double nlo;
      wave_all2<R_MAX>(hi, -lo, bhi, nlo);
      blo = -nlo;
      const double pad = 0.25 * (bhi - blo);  // (the rest of the row reaches a little further: those draws land in the end bins)
      blo -= pad;
      bhi += pad;
    }
    // bin of a draw: round((x - blo) * scale) through the 2^52 trick, one fma + one clamp; monotone in x, which is all the
    // levels need of it (NaN and +-inf never get here: the row is declined)
    const double scale = (bhi > blo && bhi - blo < 1e300) ? (double)(kQWBins - 1) / (bhi - blo) : 0.0;
    const double c0 = kMagic - blo * scale;
    const auto bin_of = [&](double x) {
      const int b = __double2loint(fma(x, scale, c0));
      return b < 0 ? 0 : (b > kQWBins - 1 ? kQWBins - 1 : b);
    };
    wave_sync();
    // ---- the trip over the row -----------------------------------------------------------------------------------------------
    double mlw = -INF;                       // running maximum of the log-weights (wave-uniform)
    double sa = 0.0, wmx = 0.0, wmn = INF;   // per lane, relative to mlw: sum, largest and smallest weight
    double xlo = INF, xhi = -INF;
    bool bad = false;
    unsigned one = 1u;
    asm volatile("" : "+v"(one));
    // ALL: every vector of the batch lies inside the row in every lane (all batches but the last one or two) -- no selects that
    // turn a slot past the end of the row into a draw that changes nothing, no branches around the two histogram atomics: a
    // quarter of the trip's vector instructions
    const auto batch = [&](const int q0, auto all_c) {
      constexpr bool ALL = decltype(all_c)::value;
      double a[kQWBatch][VEC], x[kQWBatch][VEC];
      double bm = -INF;
#pragma unroll
      for (int u = 0; u < kQWBatch; ++u) {
        unpack(tx[u], x[u]);
        unpack(tw[u], a[u]);
        const bool valid = ALL ? true : q0 + u < nval;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          bad |= valid & ((a[u][e] != a[u][e]) | !(fabs(x[u][e]) < INF));
          bm = valid ? fmax(bm, a[u][e]) : bm;
        }
      }
      if (q0 + kQWBatch < NQ) load_batch(q0 + kQWBatch);  // (the next batch flies while this one is worked on)
      bm = wave_all<R_MAX>(bm);
      if (bm > mlw) {  // (wave-uniform) a larger log-weight: bring what has been added up so far to the new maximum
        const double f = (mlw > -INF) ? wexp(mlw - bm) : 0.0;
        sa *= f;
        wmx *= f;
        wmn *= f;  // (INF stays INF)
#pragma unroll 1
        for (int b = lane; b < kQWBins; b += kWave) sm.hw[b] *= f;
        mlw = bm;
        wave_sync();
      }
      unsigned packed[kQWBatch * VEC / 2];
#pragma unroll
      for (int u = 0; u < kQWBatch; ++u) {
        const bool valid = ALL ? true : q0 + u < nval;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const double w = valid ? wexp(a[u][e] - mlw) : 0.0;
          const int b = bin_of(x[u][e]);
          if (valid) {
            atomicAdd(&sm.hw[b], w);
            atomicAdd(reinterpret_cast<unsigned*>(sm.hc) + (b >> 1), one << ((b & 1) << 4));
          }
          sa += w;
          wmx = fmax(wmx, w);
          wmn = valid ? fmin(wmn, w) : wmn;
          xlo = valid ? fmin(xlo, x[u][e]) : xlo;
          xhi = valid ? fmax(xhi, x[u][e]) : xhi;
          const unsigned bb = valid ? (unsigned)b : 0xFFFFu;  // (0xFFFF: no draw in this slot)
          const int k = u * VEC + e;
          packed[k / 2] = (k & 1) ? (packed[k / 2] | (bb << 16)) : bb;
        }
        asm volatile("" : "+v"(sa), "+v"(wmx), "+v"(wmn));  // (a vector at a time: registers)
      }
      {  // the batch's bin numbers: kQWBatch * VEC of them, two bytes each, behind the lane's earlier ones
        unsigned* dst = reinterpret_cast<unsigned*>(&sm.bins[lane * kRow + q0 * VEC]);
#pragma unroll
        for (int k = 0; k < kQWBatch * VEC / 2; k += 2) *reinterpret_cast<uint2*>(dst + k) = make_uint2(packed[k], packed[k + 1]);
      }
    };
#pragma unroll 1
    for (int q0 = 0; q0 < NQ; q0 += kQWBatch) {
      // (f64 rows only: the f32 kernel, 32 draws per lane and batch, has no registers for a second copy of the batch's code)
      if constexpr (VEC == 2) {
        if (q0 + kQWBatch <= qfull) batch(q0, std::true_type{});  // (wave-uniform)
        else batch(q0, std::false_type{});
      } else {
        batch(q0, std::false_type{});
      }
    }
    double xmax, nxmin;
    wave_all2<R_MAX>(xhi, -xlo, xmax, nxmin);
    const double xmin = -nxmin;
    if (__ballot(bad) != 0ull || !(fabs(mlw) < INF) || !(xmax > xmin) || !(scale > 0.0)) {
      decline();
      continue;
    }
    const double SA = wave_all<R_SUM>(sa);
    double WMX, nWMN;
    wave_all2<R_MAX>(wmx, -wmn, WMX, nWMN);
    {  // np.allclose(weights, weights[0]) (e_loo.py:536): np.quantile's branch is the general kernel's
      const double w0u = wexp(uniform_d((double)wr[0]) - mlw);
      const double dev = fmax(WMX - w0u, w0u - (-nWMN));
      if (dev <= kCloseAtol * SA + kCloseRtol * w0u) {
        decline();
        continue;
      }
    }
    wave_sync();
    // ---- running sums, 16 bins per lane around a scan of the lane totals (the bins are read twice: no register array) -----
    double wincl;    // cumulative weight up to and including this lane's bins
    unsigned cincl;  // ... and count
    {
      double run = 0.0;
      unsigned crun = 0u;
#pragma unroll
      for (int i = 0; i < PER; i += 2) {
        const double2 h = *reinterpret_cast<const double2*>(&sm.hw[lane * PER + i]);
        run += h.x + h.y;
      }
      unsigned cw[PER / 2];  // the lane's 16 counts, two to a word
      {
        const uint4 h0 = *reinterpret_cast<const uint4*>(&sm.hc[lane * PER]);
        const uint4 h1 = *reinterpret_cast<const uint4*>(&sm.hc[lane * PER + 8]);
        cw[0] = h0.x; cw[1] = h0.y; cw[2] = h0.z; cw[3] = h0.w;
        cw[4] = h1.x; cw[5] = h1.y; cw[6] = h1.z; cw[7] = h1.w;
#pragma unroll
        for (int i = 0; i < PER / 2; ++i) crun += (cw[i] & 0xFFFFu) + (cw[i] >> 16);
      }
      double incl = run;
      unsigned ci = crun;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const double up = __shfl_up(incl, o, kWave);
        const unsigned cu = (unsigned)__shfl_up((int)ci, o, kWave);
        if (lane >= o) {
          incl += up;
          ci += cu;
        }
      }
      double acc = incl - run;
      unsigned cacc = ci - crun;
#pragma unroll
      for (int i = 0; i < PER; i += 2) {
        const double2 h = *reinterpret_cast<const double2*>(&sm.hw[lane * PER + i]);
        const double c1 = acc + h.x, c2 = c1 + h.y;
        *reinterpret_cast<double2*>(&sm.hw[lane * PER + i]) = make_double2(c1, c2);
        acc = c2;
      }
#pragma unroll
      for (int i = 0; i < PER / 2; ++i) {
        const unsigned c1 = cacc + (cw[i] & 0xFFFFu), c2 = c1 + (cw[i] >> 16);
        cw[i] = c1 | (c2 << 16);
        cacc = c2;
      }
      *reinterpret_cast<uint4*>(&sm.hc[lane * PER]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
      *reinterpret_cast<uint4*>(&sm.hc[lane * PER + 8]) = make_uint4(cw[4], cw[5], cw[6], cw[7]);
      // (the lane's last running sum is what the look-ups compare with: the same additions in the same order)
      wincl = acc;
      cincl = cacc;
    }
    wave_sync();
    const double WTOT = lane_value(wincl, kWave - 1);  // e_loo.py:542: cumsum / sum
    bool give_up = false;
    int np = n_probs;  // (opaque per row: the loop's entry test is made here, not kept as a scalar mask from the top of the kernel)
    asm volatile("" : "+s"(np));
#pragma unroll 1
    for (int ip = 0; ip < np; ++ip) {
      const double prob = uniform_d(qp->probs[ip]);
      const double target = prob * WTOT;
      double res;
      // the bin in which the cumulative weight first reaches the target
      const unsigned long long lhit = __ballot(wincl >= target);
      if (lhit == 0ull) {
        res = xmax;  // (545-546: the total never reaches the target)
      } else {
        const int L = __ffsll((long long)lhit) - 1;
        const double cw = sm.hw[L * PER + (lane & (PER - 1))];
        const unsigned long long bhit = __ballot(cw >= target) & 0xFFFFull;
        const int tb = L * PER + (__ffsll((long long)bhit) - 1);
        const unsigned cbelow = tb > 0 ? (unsigned)sm.hc[tb - 1] : 0u;  // draws in the bins below
        int tlo = -1;                                          // the last non-empty bin below tb
        if (cbelow > 0u) {
          const unsigned long long l2 = __ballot(cincl >= cbelow);
          const int L2 = __ffsll((long long)l2) - 1;
          const unsigned cc = (unsigned)sm.hc[L2 * PER + (lane & (PER - 1))];
          const unsigned long long b2 = __ballot(cc >= cbelow) & 0xFFFFull;
          tlo = L2 * PER + (__ffsll((long long)b2) - 1);
        }
        const double base = tlo > 0 ? sm.hw[tlo - 1] : (tlo == 0 ? 0.0 : (tb > 0 ? sm.hw[tb - 1] : 0.0));
        // ---- members of the two bins: one trip over the lane's bin numbers ----------------------------------------------
        int nm = 0;
        const unsigned utb = (unsigned)tb, utlo = tlo >= 0 ? (unsigned)tlo : 0xFFFEu;
#pragma unroll 1
        for (int k0 = 0; k0 < kWaveSlots; k0 += 8) {
          const uint4 pk = *reinterpret_cast<const uint4*>(&sm.bins[lane * kRow + k0]);
          const unsigned wd[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned b = (j & 1) ? (wd[j / 2] >> 16) : (wd[j / 2] & 0xFFFFu);
            const bool hit = (b == utb) | (b == utlo);
            const unsigned long long hm = __ballot(hit);
            if (hm != 0ull) {  // (wave-uniform: most slots have no member)
              const int pos = nm + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
              if (hit && pos < kWave) {
                const int k = k0 + j;  // slot of the lane: vector k / VEC, element k % VEC
                sm.ms[pos] = VEC * (lane + kWave * (k / VEC)) + k % VEC;
              }
              nm += __popcll(hm);
            }
          }
        }
        wave_sync();
        if (nm > kWave) {
          give_up = true;
          break;
        }
        const bool mine = lane < nm;
        const int si = mine ? sm.ms[lane] : 0x7fffffff;
        const double xm = mine ? (double)xr[si] : INF;
        const double wm = mine ? wexp((double)wr[si] - mlw) : 0.0;
        double upto = base, under = base;
#pragma unroll 1
        for (int j = 0; j < nm; ++j) {
          const double xj = lane_value(xm, j), wj = lane_value(wm, j);
          upto += (xj <= xm) ? wj : 0.0;
          under += (xj < xm) ? wj : 0.0;
        }
        const bool reaches = mine && upto >= target;
        const unsigned long long rm = __ballot(reaches);
        if (rm == 0ull) {  // (rounding at the bin's edge: the general kernel decides)
          give_up = true;
          break;
        }
        const double v = wave_all<R_MIN>(reaches ? xm : INF);
        if (bin_of(v) != tb) {  // (the crossing came out in the lower bin: its predecessor was not collected)
          give_up = true;
          break;
        }
        const unsigned long long vm = __ballot(mine && xm == v);
        const double below = lane_value(under, __ffsll((long long)vm) - 1);
        // the first (lowest draw index) of the draws equal to v, its weight, and the largest draw below v
        const double sfirst = wave_all<R_MIN>((mine && xm == v) ? (double)si : 1e300);
        const unsigned long long fm = __ballot(mine && (double)si == sfirst);
        const double wfirst = lane_value(wm, __ffsll((long long)fm) - 1);
        const double prev = wave_all<R_MAX>((mine && xm < v) ? xm : -INF);
        // (e_loo.py:548-554; equal draws as the reference walks them: see the kernel above)
        if (prev == -INF) res = v;
        else if (!(below + wfirst >= target)) res = v;
        else {
          const double w1 = below / WTOT, wwi = (below + wfirst) / WTOT;
          res = prev + (v - prev) * (prob - w1) / (wwi - w1);
        }
        wave_sync();  // (the member list is rewritten by the next level)
      }
      if (lane == 0) qp->out[r * n_probs + ip] = res;
    }
    if (give_up) decline();
  }
}

}  // namespace pla
