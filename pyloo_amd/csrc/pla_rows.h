// Device code shared by the general and the fast kernels: LDS layout, row sources, exact radix
// selection, bitonic sort, the GPD fit and the reference-faithful per-row pipeline.
#pragma once

#include "../../include/pyloo_amd.h"
#include "pla_device.h"
#include "pla_kernels.h"

namespace pla {

// ------------------------------------------------------------------------------------------
// LDS carve-up (dynamic shared memory; base is 16-byte aligned, doubles first)
// ------------------------------------------------------------------------------------------
struct Smem {
  double* red;     // [16]   reduction scratch
  double* gb;      // [kMaxGrid] b_j grid of the GPD profile likelihood
  double* gl;      // [kMaxGrid] k_j, then len_scale_j
  double* part;    // [max(BLOCK, kMaxGrid)] partial log-products
  double* tx;      // [cap] tail values x (ascending after the sort); later lw+ll terms
  double* ty;      // [cap] exp(x)-exp(cut); later the smoothed log weights
  unsigned* tp;    // [cap] original draw index of each tail element
  unsigned* hist;  // [256]
  unsigned* misc;  // [8]
};

__host__ __device__ inline size_t smem_bytes(int block, int cap) {
  const int npart = block > kMaxGrid ? block : kMaxGrid;
  return sizeof(double) * (16 + 2 * kMaxGrid + npart + 2 * (size_t)cap) +
         sizeof(unsigned) * ((size_t)cap + 256 + 8);
}

template <int BLOCK>
__device__ __forceinline__ Smem carve(char* base, int cap) {
  constexpr int npart = BLOCK > kMaxGrid ? BLOCK : kMaxGrid;
  Smem s;
  double* d = reinterpret_cast<double*>(base);
  s.red = d;  d += 16;
  s.gb = d;   d += kMaxGrid;
  s.gl = d;   d += kMaxGrid;
  s.part = d; d += npart;
  s.tx = d;   d += cap;
  s.ty = d;   d += cap;
  unsigned* u = reinterpret_cast<unsigned*>(d);
  s.tp = u;   u += cap;
  s.hist = u; u += 256;
  s.misc = u;
  return s;
}

// ------------------------------------------------------------------------------------------
// Row sources.  `raw` is the log importance ratio before the max shift:  -ll in LOO mode
// (loo.py:287 passes -log_likelihood), the input itself in weights mode.
// ------------------------------------------------------------------------------------------
template <typename T, bool NEG>
struct RowGlobal {  // streams the row from global memory (L2) on every pass; any S
  const T* p;
  int64_t sd;
  int S;
  __device__ __forceinline__ double at(int s) const {
    const double v = (double)p[(int64_t)s * sd];
    return NEG ? -v : v;
  }
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int s = threadIdx.x; s < S; s += blockDim.x) f(at(s), s);
  }
};

template <typename T, bool NEG, int BLOCK, int EPT>
struct RowRegs {  // row resident in registers: EPT elements per thread, one HBM read per row
  T v[EPT];
  const T* p;
  int64_t sd;
  int S;
  __device__ __forceinline__ void load(const T* p_, int64_t sd_, int S_) {
    p = p_; sd = sd_; S = S_;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int s = threadIdx.x + i * BLOCK;
      v[i] = (s < S) ? p[(int64_t)s * sd] : T(0);
    }
  }
  __device__ __forceinline__ double at(int s) const {  // random access goes back to memory
    const double x = (double)p[(int64_t)s * sd];
    return NEG ? -x : x;
  }
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int s = threadIdx.x + i * BLOCK;
      if (s < S) f(NEG ? -(double)v[i] : (double)v[i], s);
    }
  }
};

// ------------------------------------------------------------------------------------------
// Exact selection: value of the element with 0-based rank `kth` from the top among x = raw - m.
// MSB-first radix select on order-preserving 64-bit keys, 8 bits per pass, LDS histogram.
// ------------------------------------------------------------------------------------------
template <class Row, int BLOCK>
__device__ __forceinline__ double radix_select_desc(const Row& row, double m, unsigned kth, const Smem& sm) {
  const int tid = threadIdx.x;
  uint64_t prefix = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += BLOCK) sm.hist[i] = 0;
    block_sync<BLOCK>();
    row.for_each([&](double xr, int) {
      const uint64_t key = key_of(xr - m);
      const bool act = (shift == 56) || ((key >> ((shift + 8) & 63)) == prefix);
      const unsigned dg = (unsigned)(key >> shift) & 255u;
      // leading digits are shared by almost every draw: aggregate per wave when uniform
      const uint64_t mask = __ballot(act);
      if (mask) {
        const int first = __ffsll((long long)mask) - 1;
        const unsigned d0 = (unsigned)__shfl((int)dg, first);
        const uint64_t same = __ballot(act && dg == d0);
        if (same == mask) {
          if ((tid & 63) == first) atomicAdd(&sm.hist[d0], (unsigned)__popcll(mask));
        } else if (act) {
          atomicAdd(&sm.hist[dg], 1u);
        }
      }
    });
    block_sync<BLOCK>();
    if (tid < kWave) {  // wave 0: suffix sums over the 256 bins, 4 bins per lane
      const unsigned c0 = sm.hist[4 * tid], c1 = sm.hist[4 * tid + 1];
      const unsigned c2 = sm.hist[4 * tid + 2], c3 = sm.hist[4 * tid + 3];
      const unsigned tot = c0 + c1 + c2 + c3;
      unsigned suf = tot;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const unsigned t = (unsigned)__shfl_down((int)suf, o);
        if (tid + o < kWave) suf += t;
      }
      unsigned a = suf - tot;  // draws in bins owned by higher lanes
      if (kth >= a && kth < a + tot) {
        int d;
        if (kth < a + c3) d = 3;
        else { a += c3; if (kth < a + c2) d = 2;
        else { a += c2; if (kth < a + c1) d = 1;
        else { a += c1; d = 0; } } }
        sm.misc[0] = (unsigned)(4 * tid + d);
        sm.misc[1] = kth - a;
      }
    }
    block_sync<BLOCK>();
    prefix = (prefix << 8) | sm.misc[0];
    kth = sm.misc[1];
  }
  return val_of(prefix);
}

// ascending bitonic sort of (tx, tp) pairs in LDS, P2 a power of two, ties broken by draw index
template <int BLOCK>
__device__ void bitonic_sort(double* tx, unsigned* tp, int P2) {
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P2; i += BLOCK) {
        const int q = i ^ j;
        if (q > i) {
          const double a = tx[i], b = tx[q];
          const unsigned pa = tp[i], pb = tp[q];
          const bool gt = (a > b) || (a == b && pa > pb);
          const bool up = (i & k) == 0;
          if (gt == up) { tx[i] = b; tx[q] = a; tp[i] = pb; tp[q] = pa; }
        }
      }
      block_sync<BLOCK>();
    }
  }
}

// ------------------------------------------------------------------------------------------
// Zhang-Stephens GPD fit (psis.py:163-208) on y = sm.ty[0..n) ascending.  All threads return
// the same (k, sigma).  The m_est x n matrix of log1p(-b_j*y_i) (psis.py:190) is evaluated as
// log(prod_i (1 - b_j*y_i)) with a mantissa/exponent accumulator: one log per (j, chunk).
// ------------------------------------------------------------------------------------------
struct GpdScratch {
  double* gb;    // [kMaxGrid]
  double* gl;    // [kMaxGrid]
  double* part;  // [max(BLOCK, kMaxGrid)]
  double* red;   // [16]
};

template <int BLOCK>
__device__ void gpd_fit(const double* y, int n, const GpdScratch& sm, double& k_out, double& sigma_out) {
  const int tid = threadIdx.x;
  const int mest = 30 + isqrt_i(n);                 // psis.py:184
  const double yq = y[((n + 2) >> 2) - 1];          // psis.py:187: ary[int(n/4 + 0.5) - 1]
  const double yn = y[n - 1];
  for (int j = tid; j < mest; j += BLOCK) {
    double b = 1.0 - sqrt((double)mest / ((double)(j + 1) - 0.5));  // psis.py:186
    b /= 3.0 * yq;                                                   // psis.py:187
    b += 1.0 / yn;                                                   // psis.py:188
    sm.gb[j] = b;
  }
  block_sync<BLOCK>();
  // factor range decides how often the running product must be renormalised
  const double fbig = fma(-sm.gb[0], yn, 1.0), fsmall = fma(-sm.gb[mest - 1], yn, 1.0);
  const bool wide = (fbig < 0x1p100) && (fsmall > 0x1p-100);
  const int C = (BLOCK / mest) > 0 ? (BLOCK / mest) : 1;   // chunks of the i range per j
  const int len = (n + C - 1) / C;
  for (int task = tid; task < mest * C; task += BLOCK) {
    const int j = task % mest, c = task / mest;
    const double nb = -sm.gb[j];
    int i = c * len;
    const int hi = (i + len < n) ? (i + len) : n;
    ProdAcc acc;
    acc.init();
    if (wide) {
      for (; i + 8 <= hi; i += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc.mul(fma(nb, y[i + u], 1.0));
        acc.renorm();
      }
      for (; i < hi; ++i) acc.mul(fma(nb, y[i], 1.0));
      acc.renorm();
    } else {
      for (; i < hi; ++i) { acc.mul(fma(nb, y[i], 1.0)); acc.renorm(); }
    }
    sm.part[task] = acc.log_value();
  }
  block_sync<BLOCK>();
  double lmax = -pinf();
  unsigned nanflag = 0;
  for (int j = tid; j < mest; j += BLOCK) {
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += sm.part[c * mest + j];
    const double kj = s / (double)n;                                   // psis.py:190 (.mean)
    const double ls = (double)n * (log(-(sm.gb[j] / kj)) - kj - 1.0);  // psis.py:191
    sm.gl[j] = ls;
    if (ls != ls) nanflag = 1;
    lmax = fmax(lmax, ls);
  }
  lmax = block_reduce<OpMax, BLOCK>(lmax, sm.red);
  nanflag = block_or_bits<BLOCK>(nanflag, sm.red);
  // psis.py:192: w_j = 1 / sum_l exp(ls_l - ls_j)  ==  exp(ls_j - max) / sum_l exp(ls_l - max)
  double se = 0.0;
  for (int j = tid; j < mest; j += BLOCK) se += exp(sm.gl[j] - lmax);
  se = block_reduce<OpSum, BLOCK>(se, sm.red);
  double sw = 0.0;
  for (int j = tid; j < mest; j += BLOCK) {
    const double w = nanflag ? qnan() : exp(sm.gl[j] - lmax) / se;
    if (w >= 10.0 * kEps) sw += w;                                     // psis.py:194-197
  }
  sw = block_reduce<OpSum, BLOCK>(sw, sm.red);
  double bp = 0.0;
  for (int j = tid; j < mest; j += BLOCK) {
    const double w = nanflag ? qnan() : exp(sm.gl[j] - lmax) / se;
    if (w >= 10.0 * kEps) bp += sm.gb[j] * (w / sw);                   // psis.py:198,201
  }
  const double b_post = block_reduce<OpSum, BLOCK>(bp, sm.red);        // 0 when nothing is kept
  double acc = 0.0;
  for (int i = tid; i < n; i += BLOCK) acc += log1p(-b_post * y[i]);   // psis.py:203
  const double k_post = block_reduce<OpSum, BLOCK>(acc, sm.red) / (double)n;
  sigma_out = -k_post / b_post;                                        // psis.py:205
  k_out = ((double)n * k_post + 10.0 * 0.5) / ((double)n + 10.0);      // psis.py:206
}

// ------------------------------------------------------------------------------------------
// One observation.
// ------------------------------------------------------------------------------------------
template <class Row, typename T, int BLOCK, bool LW>
__device__ __forceinline__ void process_row(const Row& row, const RowsParams& P, const Smem& sm, int64_t r) {
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  const double INF = pinf();

  // ---- 1. max, min, special values of the raw log ratios --------------------------------
  double mx = -INF, mn = INF;
  unsigned fl = 0;  // 1: NaN   2: +inf   4: -inf   (of raw)
  row.for_each([&](double xr, int) {
    if (xr != xr) fl |= 1u;
    else if (xr == INF) fl |= 2u;
    else if (xr == -INF) fl |= 4u;
    mx = fmax(mx, xr);
    mn = fmin(mn, xr);
  });
  mx = block_reduce<OpMax, BLOCK>(mx, sm.red);
  mn = block_reduce<OpMin, BLOCK>(mn, sm.red);
  fl = block_or_bits<BLOCK>(fl, sm.red);
  // NaN anywhere, or a +inf ratio (inf - inf): every shifted value is NaN/-inf, the tail is
  // empty and the normaliser is NaN (psis.py:134-158 evaluated on such a row).
  const bool bad = (fl & 3u) != 0;
  const double m = mx;  // psis.py:134

  double khat = INF;  // psis.py:142-144 default
  double xcut = 0.0, sigma = qnan();
  int n = 0;
  bool smoothed = false;

  if (!bad && P.method == PLA_PSIS) {
    // ---- 2. cutoff = max((M+1)-th largest, log(DBL_MIN))  (psis.py:135-136) --------------
    const double xk = radix_select_desc<Row, BLOCK>(row, m, (unsigned)P.tail_count, sm);
    xcut = (kLogTiny > xk) ? kLogTiny : xk;
    const double e_cut = exp(xcut);  // psis.py:138
    // ---- 3. tail = draws strictly above the cutoff (psis.py:139-141); |tail| <= M --------
    if (tid == 0) sm.misc[2] = 0;
    block_sync<BLOCK>();
    row.for_each([&](double xr, int s) {
      const double x = xr - m;
      if (x > xcut) {
        const unsigned c = atomicAdd(&sm.misc[2], 1u);
        if (c < (unsigned)P.tail_cap) { sm.tx[c] = x; sm.tp[c] = (unsigned)s; }
      }
    });
    block_sync<BLOCK>();
    n = (int)sm.misc[2];
    if (n > P.tail_cap) n = P.tail_cap;  // cannot happen (n <= M <= cap); keeps LDS in bounds
    if (n > 4) {  // psis.py:142
      const int P2 = next_pow2(n);
      for (int i = n + tid; i < P2; i += BLOCK) { sm.tx[i] = INF; sm.tp[i] = 0xffffffffu; }
      block_sync<BLOCK>();
      bitonic_sort<BLOCK>(sm.tx, sm.tp, P2);                               // psis.py:146
      for (int j = tid; j < n; j += BLOCK) sm.ty[j] = exp(sm.tx[j]) - e_cut;  // psis.py:147
      block_sync<BLOCK>();
      gpd_fit<BLOCK>(sm.ty, n, GpdScratch{sm.gb, sm.gl, sm.part, sm.red}, khat, sigma);                                  // psis.py:148
      if (isfinite(khat)) {  // psis.py:150
        block_sync<BLOCK>();
        // ---- 5. GPD quantiles at (j+0.5)/n, back to log scale, clip at 0 ----------------
        for (int j = tid; j < n; j += BLOCK) {
          const double p = ((double)j + 0.5) / (double)n;                  // psis.py:153
          double q;
          if (sigma <= 0.0) {
            q = qnan();                                                    // psis.py:214-215
          } else {
            const double l1 = log1p(-p);
            q = (fabs(khat) < kEps) ? -l1 : expm1(-khat * l1) / khat;      // psis.py:218-221
            q *= sigma;                                                    // psis.py:222
          }
          double v = log(q + e_cut);                                       // psis.py:155
          if (v > 0.0) v = 0.0;                                            // psis.py:157
          sm.ty[j] = v;
        }
        smoothed = true;
        block_sync<BLOCK>();
      }
    }
  }

  // ---- 6. normaliser  L = LSE(x')  (psis.py:158 -> utils.py:305-359) ----------------------
  double shift = 0.0;  // max of x' ; 0 unless the tail was replaced
  if (smoothed) {
    double lm = -INF;
    unsigned nanf = 0;
    for (int j = tid; j < n; j += BLOCK) {
      const double v = sm.ty[j];
      if (v != v) nanf = 1;
      lm = fmax(lm, v);
    }
    lm = block_reduce<OpMax, BLOCK>(lm, sm.red);
    nanf = block_or_bits<BLOCK>(nanf, sm.red);
    shift = nanf ? qnan() : fmax(lm, xcut);
  }
  const double maxll = -mn;  // LOO mode: raw = -ll
  double sx = 0.0, sl = 0.0;
  row.for_each([&](double xr, int) {
    const double x = xr - m;
    if (!(smoothed && x > xcut)) sx += exp(x - shift);
    if (!LW) sl += exp((-xr) - maxll);  // utils.py:349-350 for LSE(ll)
  });
  if (smoothed)
    for (int j = tid; j < n; j += BLOCK) sx += exp(sm.ty[j] - shift);
  sx = block_reduce<OpSum, BLOCK>(sx, sm.red);
  double L = log(sx) + shift;  // utils.py:352,357
  if (bad) L = qnan();

  double diag = khat;
  double tis_cut = INF;  // truncation point (TIS); +inf = no truncation
  if (P.method != PLA_PSIS) {
    if (P.method == PLA_TIS) {
      // tis.py:107-113: log_Z = LSE(x) - log S; x = min(x, log_Z + 0.5 log S); renormalise
      tis_cut = (L - log((double)S)) + 0.5 * log((double)S);
      const double sh2 = fmin(0.0, tis_cut);
      double s2 = 0.0;
      row.for_each([&](double xr, int) { s2 += exp(fmin(xr - m, tis_cut) - sh2); });
      s2 = block_reduce<OpSum, BLOCK>(s2, sm.red);
      L = log(s2) + sh2;
      if (bad) L = qnan();
    }
    // ESS = 1 / sum_s exp(lw_s)^2   (sis.py:104-105, tis.py:118-119)
    double w2 = 0.0;
    row.for_each([&](double xr, int) {
      const double w = exp(fmin(xr - m, tis_cut) - L);
      w2 += w * w;
    });
    w2 = block_reduce<OpSum, BLOCK>(w2, sm.red);
    diag = 1.0 / w2;
  }

  if constexpr (LW) {
    // ---- weights mode: write lw (base.py:160-166 outputs) -------------------------------
    T* out = reinterpret_cast<T*>(P.lw_out) + r * (int64_t)S;
    row.for_each([&](double xr, int s) {
      const double x = xr - m;
      if (!(smoothed && x > xcut)) out[s] = (T)(fmin(x, tis_cut) - L);
    });
    if (smoothed)
      for (int j = tid; j < n; j += BLOCK) out[sm.tp[j]] = (T)(sm.ty[j] - L);  // psis.py:156
    if (tid == 0 && P.diag) P.diag[r] = diag;
  } else {
    // ---- LOO mode: loo_i = LSE_s(lw_s + ll_s), lppd_i = LSE_s(ll_s) - log S -------------
    sl = block_reduce<OpSum, BLOCK>(sl, sm.red);
    const double lppd = log(sl) + (maxll - log((double)S));  // utils.py:352-357 with b_inv = S
    const double A = (-m) - L;  // lw_s + ll_s for every draw whose weight was not replaced
    double loo;
    if (smoothed) {
      double tm = -INF;
      for (int j = tid; j < n; j += BLOCK) {
        const double llj = -row.at((int)sm.tp[j]);
        const double t = (sm.ty[j] - L) + llj;  // loo.py:289 on a smoothed draw
        sm.tx[j] = t;
        tm = fmax(tm, t);
      }
      tm = block_reduce<OpMax, BLOCK>(tm, sm.red);  // (also orders the tx writes)
      const double top = fmax(A, tm);
      double acc = 0.0;
      for (int j = tid; j < n; j += BLOCK) acc += exp(sm.tx[j] - top);
      acc = block_reduce<OpSum, BLOCK>(acc, sm.red);
      loo = top + log((double)(S - n) * exp(A - top) + acc);
      if (L != L) loo = qnan();
    } else if (P.method == PLA_TIS) {
      // lw_s + ll_s = min(x_s, cut) - x_s - m - L
      double acc = 0.0;
      row.for_each([&](double xr, int) { acc += exp(fmin(xr - m, tis_cut) - (xr - m)); });
      acc = block_reduce<OpSum, BLOCK>(acc, sm.red);
      loo = A + log(acc);
    } else {
      loo = A + log((double)S);
    }
    if (bad || (fl & 4u)) loo = qnan();  // ll = +inf: lw + ll = -inf + inf (loo.py:289)
    if (tid == 0) {
      if (P.diag) P.diag[r] = diag;
      if (P.loo_i) P.loo_i[r] = P.scale_value * loo;  // loo.py:319
      if (P.lppd_i) P.lppd_i[r] = lppd;
    }
  }
  block_sync<BLOCK>();  // LDS is reused by the next row
}

}  // namespace pla
