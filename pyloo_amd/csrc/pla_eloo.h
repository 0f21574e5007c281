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

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void e_loo_rows_kernel(ELooParams P) {
  __shared__ double red[(BLOCK / kWave) * 12];
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  const int n_tail = S < P.tail_len ? S : P.tail_len;
  const bool own_lr = P.lr != P.lw;
  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
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
    if (tid == 0) {
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
      const bool skip_x = (xdev <= kCloseAtol + kCloseRtol * fabs(x0) && !(flags & 4u)) || (other_x == 0.0 && xmn != xmx && !(flags & 4u)) ||
                          (flags & (4u | 8u)) != 0u;
      const bool skip_q = (qdev <= kCloseAtol + kCloseRtol * fabs(q0) && !(flags & 4u)) || (other_q == 0.0 && qmn != qmx && !(flags & 4u)) ||
                          (flags & (4u | 16u)) != 0u;
      if (P.k_mean) P.k_mean[r] = k_of(skip_x, c1l, c1r);
      if (P.k_var) P.k_var[r] = k_of(skip_q, c2l, c2r);
      if (P.k_none) P.k_none[r] = k_r;
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
// Equal draws are treated as one draw carrying their combined weight (the reference orders ties by an unstable argsort;
// the result differs only when `prob` is crossed inside a group of equal draws with unequal weights).
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
};

// smallest key whose cumulative mass (sum of `mass(s)` over draws with key <= it) reaches `target`; *below = mass strictly
// below that key, *at = mass at it.  Returns false when the total never reaches the target.
template <int BLOCK, class KeyAt, class MassAt>
__device__ __forceinline__ bool mass_select(const int S, KeyAt key_at, MassAt mass_at, const double target, double* hist, double* red,
                                            uint64_t* key_out, double* below, double* at) {
  const int tid = threadIdx.x;
  uint64_t prefix = 0;
  double base = 0.0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += BLOCK) hist[i] = 0.0;
    __syncthreads();
    for (int s = tid; s < S; s += BLOCK) {
      const uint64_t k = key_at(s);
      if (shift == 56 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], mass_at(s));
    }
    __syncthreads();
    // one thread walks the 256 bins (the walk is short and the order of the additions is then fixed)
    if (tid == 0) {
      double cum = base;
      int d = 0;
      for (; d < 256; ++d) {
        if (cum + hist[d] >= target) break;
        cum += hist[d];
      }
      red[0] = cum;
      red[1] = (double)d;
      red[2] = d < 256 ? hist[d] : 0.0;
    }
    __syncthreads();
    const int d = (int)red[1];
    base = red[0];
    const double here = red[2];
    __syncthreads();
    if (d >= 256) return false;
    prefix = (prefix << 8) | (uint64_t)d;
    if (shift == 0) {
      *key_out = prefix;
      *below = base;
      *at = here;
    }
  }
  return true;
}

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void e_loo_quantile_kernel(EQuantParams P) {
  __shared__ double hist[256];
  __shared__ double red[BLOCK / kWave > 4 ? BLOCK / kWave : 4];
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const T* xr = reinterpret_cast<const T*>(P.x) + r * P.stride_obs;
    const T* wr = reinterpret_cast<const T*>(P.lw) + r * P.stride_obs;
    const auto xat = [&](int s) { return (double)xr[(int64_t)s * P.stride_draw]; };
    const auto key_at = [&](int s) { return key_of(xat(s)); };
    // normalised weights w = exp(lw - logsumexp(lw)) (e_loo.py:557-559, 473): maximum, sum, and whether they are all close
    double mlw = -pinf(), xmax = -pinf(), xmin = pinf();
    unsigned nanw = 0;
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
    double sa = 0.0;
    for (int s = tid; s < S; s += BLOCK) sa += exp((double)wr[(int64_t)s * P.stride_draw] - mlw);
    sa = block_reduce<OpSum, BLOCK>(sa, red);
    const double w0 = exp((double)wr[0] - mlw) / sa;
    double dev = 0.0;
    for (int s = tid; s < S; s += BLOCK) dev = fmax(dev, fabs(exp((double)wr[(int64_t)s * P.stride_draw] - mlw) / sa - w0));
    dev = block_reduce<OpMax, BLOCK>(dev, red);
    const bool flat = !nanw && dev <= kCloseAtol + kCloseRtol * fabs(w0);                 // e_loo.py:536
    const auto wat = [&](int s) { return exp((double)wr[(int64_t)s * P.stride_draw] - mlw) / sa; };
    for (int ip = 0; ip < P.n_probs; ++ip) {
      const double prob = P.probs[ip];
      double res;
      uint64_t kv = 0;
      double below = 0.0, at = 0.0;
      if (flat) {
        // np.quantile(x, prob), method "linear": virtual index (S - 1) prob between the order statistics lo and lo + 1
        const double virt = (double)(S - 1) * prob;
        const double lo = floor(virt), t = virt - lo;
        const auto one = [](int) { return 1.0; };
        mass_select<BLOCK>(S, key_at, one, lo + 1.0, hist, red, &kv, &below, &at);
        const double a = val_of(kv);
        double b = a;
        if (below + at < lo + 2.0 && lo + 1.0 < (double)S) {  // the next order statistic is the next distinct value
          double nxt = pinf();
          for (int s = tid; s < S; s += BLOCK) {
            const double x = xat(s);
            if (key_at(s) > kv) nxt = fmin(nxt, x);
          }
          b = block_reduce<OpMin, BLOCK>(nxt, red);
        }
        const double diff = b - a;
        res = (t >= 0.5) ? b - diff * (1.0 - t) : a + diff * t;                           // numpy's _lerp
        if (t == 0.0) res = a;
      } else {
        const auto wmass = [&](int s) { return wat(s); };
        double wtot = 0.0;
        for (int s = tid; s < S; s += BLOCK) wtot += wat(s);
        wtot = block_reduce<OpSum, BLOCK>(wtot, red);                                     // e_loo.py:542: cumsum / sum
        const bool found = mass_select<BLOCK>(S, key_at, wmass, prob * wtot, hist, red, &kv, &below, &at);
        if (!found) {
          res = xmax;                                                                     // 545-546
        } else {
          const double v = val_of(kv);
          double prev = -pinf();
          for (int s = tid; s < S; s += BLOCK)
            if (key_at(s) < kv) prev = fmax(prev, xat(s));
          prev = block_reduce<OpMax, BLOCK>(prev, red);
          if (below == 0.0 && prev == -pinf()) res = v;                                   // wi == 0: 548-550
          else {
            const double w1 = below / wtot, wwi = (below + at) / wtot;                    // 552
            res = prev + (v - prev) * (prob - w1) / (wwi - w1);                           // 554
          }
        }
      }
      if (tid == 0) P.out[r * P.n_probs + ip] = res;
      __syncthreads();
    }
  }
}

}  // namespace pla
