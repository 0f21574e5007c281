// Fast LOO kernel: one 256-thread workgroup per observation, the row lives in registers (one
// HBM read per row), every pass after the load runs out of registers / LDS.
//
//   load 16 values/thread (16-byte loads)          -> max, min, non-finite count, min of the
//                                                     per-thread maxima (coarse threshold t1)
//   ONE sweep over the registers                   -> e^x and e^(ll-max ll) for all S draws from a
//                                                     single range reduction (the two exponents
//                                                     are -u and u-R), LDS histogram of the draws
//                                                     above t1 in 1024 linear bins
//   suffix scan of the bins                        -> boundary bin holding the (M+1)-th largest
//   second sweep                                   -> draws at/above the boundary bin to LDS,
//                                                     grouped by bin; exact rank inside each bin
//   GPD fit + smoothing on the <= M tail in LDS    -> khat, sum of smoothed weights
//   loo_i, lppd_i from the sums (no extra pass)
//
// Rows the shortcuts cannot represent exactly as the reference computes them (non-finite
// entries, a range above 700 nats where the log(DBL_MIN) floor of psis.py:136 can bind, too many
// candidates in the boundary bins, non-finite results) are appended to a list and recomputed by
// the general kernel (pla_rows.h) -- same results, just slower.
#pragma once

#include "pla_rows.h"

namespace pla {

constexpr int kFastBlock = 256;
constexpr int kFastBins = 4 * kFastBlock;  // 4 bins per thread in the scan
constexpr int kFastCap = 512;              // candidates kept in LDS (>= M + boundary bin)
constexpr double kFastMaxRange = 700.0;    // nats; keeps every exponential normal

struct FastSmem {
  double* red;     // [40]
  double* gb;      // [kMaxGrid]
  double* gl;      // [kMaxGrid]
  double* part;    // [kFastBlock]
  double* sa;      // [cap] staging by bin, later y ascending
  double* sb;      // [cap] candidates sorted descending
  unsigned* pa;    // [cap]
  unsigned* pb;    // [cap]
  unsigned* hist;  // [kFastBins]
  unsigned* start; // [kFastBins]
  unsigned* misc;  // [16]
};

__host__ __device__ inline size_t fast_smem_bytes() {
  return sizeof(double) * (40 + 2 * kMaxGrid + kFastBlock + 2 * kFastCap) +
         sizeof(unsigned) * (2 * kFastCap + 2 * kFastBins + 16);
}

__device__ __forceinline__ FastSmem fast_carve(char* base) {
  FastSmem s;
  double* d = reinterpret_cast<double*>(base);
  s.red = d;  d += 40;
  s.gb = d;   d += kMaxGrid;
  s.gl = d;   d += kMaxGrid;
  s.part = d; d += kFastBlock;
  s.sa = d;   d += kFastCap;
  s.sb = d;   d += kFastCap;
  unsigned* u = reinterpret_cast<unsigned*>(d);
  s.pa = u;    u += kFastCap;
  s.pb = u;    u += kFastCap;
  s.hist = u;  u += kFastBins;
  s.start = u; u += kFastBins;
  s.misc = u;
  return s;
}

// ---- several sums / maxima / minima in one round trip through LDS (2 barriers) ------------
enum RedOp { R_SUM, R_MAX, R_MIN };
template <RedOp OP>
__device__ __forceinline__ double red_apply(double a, double b) {
  if constexpr (OP == R_SUM) return a + b;
  else if constexpr (OP == R_MAX) return fmax(a, b);
  else return fmin(a, b);
}
template <int I, RedOp OP, RedOp... REST>
struct RedWalk {
  template <int N>
  __device__ static __forceinline__ void wave(double (&v)[N]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[I] = red_apply<OP>(v[I], __shfl_xor(v[I], o));
    if constexpr (sizeof...(REST) > 0) RedWalk<I + 1, REST...>::wave(v);
  }
  template <int N, int NW>
  __device__ static __forceinline__ void combine(double (&v)[N], const double* scratch) {
    v[I] = scratch[I];
#pragma unroll
    for (int w = 1; w < NW; ++w) v[I] = red_apply<OP>(v[I], scratch[w * N + I]);
    if constexpr (sizeof...(REST) > 0) RedWalk<I + 1, REST...>::template combine<N, NW>(v, scratch);
  }
};
template <int BLOCK, RedOp... OPS>
__device__ __forceinline__ void block_reduce_multi(double (&v)[sizeof...(OPS)], double* scratch) {
  constexpr int N = sizeof...(OPS);
  constexpr int NW = BLOCK / kWave;
  RedWalk<0, OPS...>::wave(v);
  if constexpr (NW > 1) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) scratch[w * N + i] = v[i];
    }
    __syncthreads();
    RedWalk<0, OPS...>::template combine<N, NW>(v, scratch);
  }
}

// ---- e^x and e^(-x-R) from one range reduction ---------------------------------------------
// x in [-R, 0], R < 700.  x = k ln2 + r, |r| <= ln2/2:  e^x = 2^k P(r),  e^(-x-R) = 2^(-k-kR) P(-r) cR
// with e^-R = cR 2^-kR.  P is the degree-13 Taylor polynomial (|error| < 5e-18 on the interval),
// evaluated as even + odd parts so P(r) and P(-r) share every multiply.
struct ExpPair {
  double cR;
  int kR;
  __device__ __forceinline__ void init(double R) {
    const double kf = rint(R * 1.4426950408889634);
    double r = fma(kf, -6.93147180369123816490e-01, R);
    r = fma(kf, -1.90821492927058770002e-10, r);
    cR = exp(-r);
    kR = (int)kf;
  }
  __device__ __forceinline__ void eval(double x, double& e1, double& e2) const {
    const double kf = rint(x * 1.4426950408889634);
    double r = fma(kf, -6.93147180369123816490e-01, x);
    r = fma(kf, -1.90821492927058770002e-10, r);
    const double r2 = r * r;
    double E = 2.08767569878680989792e-09;            // 1/12!
    E = fma(E, r2, 2.75573192239858906526e-07);       // 1/10!
    E = fma(E, r2, 2.48015873015873015873e-05);       // 1/8!
    E = fma(E, r2, 1.38888888888888888889e-03);       // 1/6!
    E = fma(E, r2, 4.16666666666666666667e-02);       // 1/4!
    E = fma(E, r2, 0.5);
    E = fma(E, r2, 1.0);
    double O = 1.60590438368216145994e-10;            // 1/13!
    O = fma(O, r2, 2.50521083854417187751e-08);       // 1/11!
    O = fma(O, r2, 2.75573192239858906526e-06);       // 1/9!
    O = fma(O, r2, 1.98412698412698412698e-04);       // 1/7!
    O = fma(O, r2, 8.33333333333333333333e-03);       // 1/5!
    O = fma(O, r2, 1.66666666666666666667e-01);       // 1/3!
    O = fma(O, r2, 1.0);
    const double rO = r * O;
    const int k = (int)kf;
    e1 = ldexp(E + rO, k);
    e2 = ldexp((E - rO) * cR, -k - kR);
  }
};

struct FastParams {
  int gsz;                        // slots per threshold group (power of two, <= EPT)
  unsigned* slow_list;            // [n_obs] rows for the general kernel
  unsigned long long* counters;   // [0] = number of rows in slow_list
  int debug_skip;                 // phase-ablation bits for profiling (0 in production)
  double* dbg;                    // debug dump (null in production)
  const double* l1_table;         // [M] log1p(-(j+0.5)/M), host-computed (wave kernel)
  double log_S;                   // log(n_draws)
  const double* b_grid;           // [64] 1 - sqrt(m_est/(j+0.5)) for m_est = mest_M (psis.py:186)
  int mest_M;                     // 30 + isqrt(M)
};

template <int BLOCK>
__device__ __forceinline__ int fast_bin(double x, double t1, double scale) {
  const int b = (int)((x - t1) * scale);
  return b < kFastBins - 1 ? b : kFastBins - 1;
}

template <typename T, int EPT, int VEC>
__global__ __launch_bounds__(kFastBlock, 2) void fast_loo_kernel(RowsParams P, FastParams F) {
  constexpr int BLOCK = kFastBlock;
  static_assert(EPT % VEC == 0, "EPT must be a multiple of the vector width");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const FastSmem sm = fast_carve(smem_raw);
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  const int M = P.tail_count;
  const double INF = pinf();
  typedef T VT __attribute__((ext_vector_type(VEC)));

  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const T* rp = reinterpret_cast<const T*>(P.in) + r * P.stride_obs;
    // ---- load: slot i holds draw  VEC*(tid + BLOCK*(i/VEC)) + i%VEC --------------------------
    T v[EPT];
#pragma unroll
    for (int q = 0; q < EPT / VEC; ++q) {
      const int s0 = VEC * (tid + BLOCK * q);
      if (s0 < S) {  // S % VEC == 0 is guaranteed by the launcher
        const VT t = *reinterpret_cast<const VT*>(rp + s0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q * VEC + e] = t[e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q * VEC + e] = T(0);
      }
    }
    // ---- 1. row statistics ------------------------------------------------------------------
    double st[4] = {-INF, INF, INF, 0.0};  // max raw, min raw, min of group maxima, #non-finite
    {
      double gcur = -INF;
      bool ghas = false;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int s = VEC * (tid + BLOCK * (i / VEC)) + (i % VEC);
        if (s < S) {
          const double raw = -(double)v[i];
          st[0] = fmax(st[0], raw);
          st[1] = fmin(st[1], raw);
          if (!(fabs(raw) <= 1.7976931348623157e308)) st[3] += 1.0;
          gcur = fmax(gcur, raw);
          ghas = true;
        }
        if (((i + 1) & (F.gsz - 1)) == 0) {
          if (ghas) st[2] = fmin(st[2], gcur);
          gcur = -INF;
          ghas = false;
        }
      }
    }
    block_reduce_multi<BLOCK, R_MAX, R_MIN, R_MIN, R_SUM>(st, sm.red);
    const double m = st[0], mn = st[1];
    const double R = m - mn;
    const double t1 = st[2] - m;  // <= 0; at least (#groups) >= M+1 draws have x >= t1
    bool slow = (st[3] != 0.0) || !(R < kFastMaxRange) || !(t1 < 0.0);
    double khat = INF, loo = 0.0, lppd = 0.0;
    if (!slow) {
      const double scale = (double)kFastBins / (-t1);
      for (int i = tid; i < kFastBins; i += BLOCK) sm.hist[i] = 0;
      __syncthreads();
      // ---- 2. one sweep: both exponentials of every draw + histogram of the candidates -------
      ExpPair ep;
      ep.init(R);
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int s = VEC * (tid + BLOCK * (i / VEC)) + (i % VEC);
        if (s < S) {
          const double x = (-(double)v[i]) - m;  // psis.py:134
          double e1 = 0.0, e2 = 0.0;
          if (!(F.debug_skip & 1)) ep.eval(x, e1, e2);
          s1 += e1;
          s2 += e2;
          if (x >= t1 && !(F.debug_skip & 2)) atomicAdd(&sm.hist[fast_bin<BLOCK>(x, t1, scale)], 1u);
        }
      }
      __syncthreads();
      // ---- 3. suffix scan: start[b] = #draws in bins above b; boundary bin holds rank M -------
      {
        const unsigned c0 = sm.hist[4 * tid], c1 = sm.hist[4 * tid + 1];
        const unsigned c2 = sm.hist[4 * tid + 2], c3 = sm.hist[4 * tid + 3];
        const unsigned tot = c0 + c1 + c2 + c3;
        unsigned suf = tot;
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const unsigned t = (unsigned)__shfl_down((int)suf, o);
          if (lane + o < kWave) suf += t;
        }
        if (lane == 0) sm.misc[4 + w] = suf;
        __syncthreads();
        unsigned hi = 0;
#pragma unroll
        for (int ww = 0; ww < BLOCK / kWave; ++ww)
          if (ww > w) hi += sm.misc[4 + ww];
        unsigned a = hi + suf - tot;
        sm.start[4 * tid + 3] = a;
        sm.start[4 * tid + 2] = a + c3;
        sm.start[4 * tid + 1] = a + c3 + c2;
        sm.start[4 * tid] = a + c3 + c2 + c1;
        const unsigned kth = (unsigned)M;
        if (kth >= a && kth < a + tot) {
          int d;
          unsigned cnt;
          if (kth < a + c3) { d = 3; cnt = c3; }
          else { a += c3; if (kth < a + c2) { d = 2; cnt = c2; }
          else { a += c2; if (kth < a + c1) { d = 1; cnt = c1; }
          else { a += c1; d = 0; cnt = c0; } } }
          sm.misc[0] = (unsigned)(4 * tid + d);  // boundary bin
          sm.misc[1] = a + cnt;                  // candidates to keep (everything >= boundary bin)
        }
        __syncthreads();
      }
      const int bstar = (int)sm.misc[0];
      const int C1 = (int)sm.misc[1];
      if (F.debug_skip & 4) {
        loo = s1; lppd = s2;
      } else if (C1 > kFastCap) {
        slow = true;
      } else {
        // ---- 4. candidates -> LDS, grouped by bin (descending bins) -----------------------------
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const int s = VEC * (tid + BLOCK * (i / VEC)) + (i % VEC);
          if (s < S) {
            const double x = (-(double)v[i]) - m;
            if (x >= t1) {
              const int b = fast_bin<BLOCK>(x, t1, scale);
              if (b >= bstar) {
                const unsigned slot = sm.start[b] + (atomicSub(&sm.hist[b], 1u) - 1u);
                sm.sa[slot] = x;
                sm.pa[slot] = (unsigned)s;
              }
            }
          }
        }
        __syncthreads();
        // ---- 5. exact descending rank inside each bin (ties: later draw first) -----------------
        for (int c = tid; c < C1; c += BLOCK) {
          const double x = sm.sa[c];
          const unsigned p = sm.pa[c];
          const int b = fast_bin<BLOCK>(x, t1, scale);
          const int lo = (F.debug_skip & 16) ? c : (int)sm.start[b];
          const int hi = (F.debug_skip & 16) ? c : ((b > 0) ? (int)sm.start[b - 1] : C1);
          int cnt = 0;
          for (int c2 = lo; c2 < hi; ++c2) {
            const double x2 = sm.sa[c2];
            cnt += (x2 > x || (x2 == x && sm.pa[c2] > p)) ? 1 : 0;
          }
          sm.sb[lo + cnt] = x;
          sm.pb[lo + cnt] = p;
        }
        __syncthreads();
        // ---- cutoff (psis.py:135-141); R < 700 means the log(DBL_MIN) floor cannot bind ---------
        const double xcut = sm.sb[M];
        int n = M;
        while (n > 0 && sm.sb[n - 1] == xcut) --n;  // ties at the cutoff leave the tail
        const double e_cut = exp(xcut);
        double sums[5] = {s1, s2, 0.0, 0.0, 0.0};  // sum e^x, sum e^(ll-max), sum w', sum w'/e, sum e (tail)
        bool smoothed = false;
        double sigma = qnan();
        if (n > 4 && !(F.debug_skip & 8)) {
          for (int j = tid; j < n; j += BLOCK) sm.sa[j] = exp(sm.sb[n - 1 - j]) - e_cut;  // psis.py:147
          __syncthreads();
          GpdScratch gs{sm.gb, sm.gl, sm.part, sm.red};
          gpd_fit<BLOCK>(sm.sa, n, gs, khat, sigma);
          if (isfinite(khat)) {
            smoothed = true;
            for (int j = tid; j < n; j += BLOCK) {
              const double p = ((double)j + 0.5) / (double)n;
              double q;
              if (sigma <= 0.0) {
                q = qnan();
              } else {
                const double l1 = log1p(-p);
                q = (fabs(khat) < kEps) ? -l1 : expm1(-khat * l1) / khat;
                q *= sigma;
              }
              double wj = q + e_cut;       // exp(log(q + e_cut)), psis.py:155
              if (wj > 1.0) wj = 1.0;      // psis.py:157
              const double ej = sm.sa[j] + e_cut;
              sums[2] += wj;
              sums[3] += wj / ej;
              sums[4] += ej;
            }
          }
        }
        block_reduce_multi<BLOCK, R_SUM, R_SUM, R_SUM, R_SUM, R_SUM>(sums, sm.red);
        const double total = smoothed ? (sums[0] - sums[4]) + sums[2] : sums[0];
        const double L = log(total);                       // psis.py:158
        const double A = (-m) - L;
        loo = smoothed ? A + log((double)(S - n) + sums[3]) : A + log((double)S);
        lppd = log(sums[1]) + ((-mn) - log((double)S));   // loo.py:329-337
        if ((!(total > 1e-280) || !isfinite(loo) || !isfinite(lppd)) && !F.debug_skip) slow = true;
      }
    }
    if (tid == 0) {
      if (slow) {
        const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
        F.slow_list[idx] = (unsigned)r;
      } else {
        if (P.diag) P.diag[r] = khat;
        if (P.loo_i) P.loo_i[r] = P.scale_value * loo;
        if (P.lppd_i) P.lppd_i[r] = lppd;
      }
    }
    __syncthreads();
  }
}

}  // namespace pla
