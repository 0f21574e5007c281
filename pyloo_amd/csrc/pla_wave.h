// Wave-per-observation LOO kernel (the production fast path for S <= 4096 draws).
//
// One 64-lane wavefront owns one observation: the row of S draws sits in its registers
// (64 slots per lane, loaded once with 16-byte loads), and every later step is done by that
// wave alone with cross-lane shuffles and a private LDS scratch -- there is NO workgroup
// barrier anywhere, so the 8 waves resident on a CU run their phases (HBM load, exp sweep,
// selection, GPD fit) completely decoupled and cover each other's latencies.
//
//   stats     max / min / non-finite count / min over groups of the group maxima (threshold t1:
//             at least #groups >= M+1 draws lie at or above it)
//   sweep     for every draw: e^x and e^(ll - max ll) from ONE range reduction with a 32-entry
//             2^(j/32) table and a degree-6 polynomial (the exponents are x and -x-R);
//             draws >= t1 are counted in a 1024-bin linear LDS histogram
//   select    suffix scan -> boundary bin of rank M; candidates at/above it are scattered to
//             LDS grouped by bin and ranked exactly inside their bin
//   fit       Zhang-Stephens GPD fit, one lane per grid point b_j, log(prod) instead of sum(log1p)
//   smooth    GPD quantiles, sums of the smoothed weights; loo_i / lppd_i from the sums
//
// Rows the shortcuts cannot reproduce exactly as the reference computes them are appended to a
// list and recomputed by the general kernel (pla_rows.h).
#pragma once

#include "pla_fast.h"

namespace pla {

constexpr int kWaveSlots = 64;   // register slots per lane -> S <= 4096
constexpr int kWaveBins = 1024;
constexpr int kWaveCap = 320;    // candidates kept in LDS; needs M + boundary-bin extras
constexpr int kWaveMaxTail = 250;

__device__ __forceinline__ double uniform_d(double v) {  // value is wave-uniform: move to SGPRs
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

template <RedOp OP>
__device__ __forceinline__ double wave_all(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = red_apply<OP>(v, __shfl_xor(v, o));
  return uniform_d(v);
}

__device__ __forceinline__ int wave_bin(double x, double t1, double scale) {
  const int b = (int)((x - t1) * scale);
  return b < kWaveBins - 1 ? b : kWaveBins - 1;
}

struct WaveSmem {
  unsigned hist[kWaveBins];
  unsigned start[kWaveBins];
  double sa[kWaveCap];
  double sb[kWaveCap];
  double tab[64];  // [0,32): 2^(j/32)   [32,64): 2^(-j/32) * cR
};

template <typename T, int VEC>
__global__ __launch_bounds__(kWave, 2) void wave_loo_kernel(RowsParams P, FastParams F) {
  constexpr int EPT = kWaveSlots;
  __shared__ WaveSmem sm;
  const int lane = threadIdx.x;
  const int S = P.n_draws;
  const int M = P.tail_count;
  const double INF = pinf();
  typedef T VT __attribute__((ext_vector_type(VEC)));

  // 2^(j/32), j = 0..31 (exact to rounding; exp2 is correctly rounded enough at these points)
  if (lane < 32) sm.tab[lane] = exp2((double)lane * 0.03125);
  __syncthreads();

  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const T* rp = reinterpret_cast<const T*>(P.in) + r * P.stride_obs;
    // ---- load: slot i holds draw  VEC*(lane + 64*(i/VEC)) + i%VEC ------------------------------
    T v[EPT];
#pragma unroll
    for (int q = 0; q < EPT / VEC; ++q) {
      const int s0 = VEC * (lane + kWave * q);
      if (s0 < S) {
        const VT t = __builtin_nontemporal_load(reinterpret_cast<const VT*>(rp + s0));
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q * VEC + e] = t[e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q * VEC + e] = T(0);
      }
    }
    // ---- 1. row statistics ----------------------------------------------------------------------
    double mx = -INF, mn = INF, gmin = INF, nbad = 0.0;
    {
      double gcur = -INF;
      bool ghas = false;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int s = VEC * (lane + kWave * (i / VEC)) + (i % VEC);
        if (s < S) {
          const double raw = -(double)v[i];
          mx = fmax(mx, raw);
          mn = fmin(mn, raw);
          if (!(fabs(raw) <= 1.7976931348623157e308)) nbad += 1.0;
          gcur = fmax(gcur, raw);
          ghas = true;
        }
        if (((i + 1) & (F.gsz - 1)) == 0) {
          if (ghas) gmin = fmin(gmin, gcur);
          gcur = -INF;
          ghas = false;
        }
      }
    }
    const double m = wave_all<R_MAX>(mx);
    mn = wave_all<R_MIN>(mn);
    gmin = wave_all<R_MIN>(gmin);
    nbad = wave_all<R_SUM>(nbad);
    const double R = m - mn;
    const double t1 = gmin - m;
    bool slow = (nbad != 0.0) || !(R < kFastMaxRange) || !(t1 < 0.0);
    double khat = INF, loo = 0.0, lppd = 0.0;
    if (!slow) {
      const double scale = (double)kWaveBins / (-t1);
      // per-row pieces of the second exponential: e^-R = cR * 2^-kR
      double cR;
      int kR;
      {
        const double kf = rint(R * 1.4426950408889634);
        double rr = fma(kf, -6.93147180369123816490e-01, R);
        rr = fma(kf, -1.90821492927058770002e-10, rr);
        cR = exp(-rr);
        kR = (int)kf;
      }
      __syncthreads();  // previous row is done with the tables / histogram
      if (lane < 32) sm.tab[32 + lane] = cR / sm.tab[lane];
#pragma unroll
      for (int i = 0; i < kWaveBins / kWave; ++i) sm.hist[lane + kWave * i] = 0;
      __syncthreads();
      // ---- 2. sweep: both exponentials of every draw + histogram of the candidates -----------
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int s = VEC * (lane + kWave * (i / VEC)) + (i % VEC);
        if (s < S) {
          const double x = (-(double)v[i]) - m;  // psis.py:134
          if (!(F.debug_skip & 1)) {
            const double kf = rint(x * 46.16624130844683);  // 32 / ln 2
            double rr = fma(kf, -2.16608493865351192653e-02, x);   // ln2_hi / 32
            rr = fma(kf, -5.96317165397058656257e-12, rr);          // ln2_lo / 32
            const int k = (int)kf;
            const int j = k & 31;
            const int e = k >> 5;
            const double tj = sm.tab[j], ij = sm.tab[32 + j];
            const double r2 = rr * rr;
            double E = fma(1.38888888888888888889e-03, r2, 4.16666666666666666667e-02);
            E = fma(E, r2, 0.5);
            E = fma(E, r2, 1.0);
            double O = fma(8.33333333333333333333e-03, r2, 1.66666666666666666667e-01);
            O = fma(O, r2, 1.0);
            const double rO = rr * O;
            s1 += ldexp(tj * (E + rO), e);
            s2 += ldexp(ij * (E - rO), -e - kR);
          }
          if (x >= t1 && !(F.debug_skip & 2)) atomicAdd(&sm.hist[wave_bin(x, t1, scale)], 1u);
        }
      }
      __syncthreads();
      if (F.debug_skip & 4) {
        loo = s1;
        lppd = s2;
      } else {
        // ---- 3. suffix scan (16 bins per lane): start[b] = #draws in bins above b ---------------
        int bstar = 0, C1 = 0;
        {
          unsigned c[16];
          unsigned tot = 0;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            c[i] = sm.hist[16 * lane + i];
            tot += c[i];
          }
          unsigned suf = tot;
#pragma unroll
          for (int o = 1; o < kWave; o <<= 1) {
            const unsigned t = (unsigned)__shfl_down((int)suf, o);
            if (lane + o < kWave) suf += t;
          }
          unsigned a = suf - tot;  // draws in bins owned by higher lanes
          int fb = -1, fc = 0;
#pragma unroll
          for (int i = 15; i >= 0; --i) {
            sm.start[16 * lane + i] = a;
            if (fb < 0 && (unsigned)M >= a && (unsigned)M < a + c[i]) {
              fb = 16 * lane + i;
              fc = (int)(a + c[i]);
            }
            a += c[i];
          }
          // exactly one lane found the boundary bin
          const unsigned long long who = __ballot(fb >= 0);
          const int src = __ffsll((long long)who) - 1;
          bstar = __shfl(fb, src);
          C1 = __shfl(fc, src);
        }
        __syncthreads();
        if (C1 > kWaveCap) {
          slow = true;
        } else {
          // ---- 4. candidates -> LDS grouped by bin (descending bins) -----------------------------
#pragma unroll
          for (int i = 0; i < EPT; ++i) {
            const int s = VEC * (lane + kWave * (i / VEC)) + (i % VEC);
            if (s < S) {
              const double x = (-(double)v[i]) - m;
              if (x >= t1) {
                const int b = wave_bin(x, t1, scale);
                if (b >= bstar) {
                  const unsigned slot = sm.start[b] + (atomicSub(&sm.hist[b], 1u) - 1u);
                  sm.sa[slot] = x;
                }
              }
            }
          }
          __syncthreads();
          // ---- 5. exact descending rank inside each bin (ties: arbitrary, the sums do not care) --
          for (int c = lane; c < C1; c += kWave) {
            const double x = sm.sa[c];
            const int b = wave_bin(x, t1, scale);
            const int lo = (int)sm.start[b];
            const int hi = (b > 0) ? (int)sm.start[b - 1] : C1;
            int cnt = 0;
            for (int c2 = lo; c2 < hi; ++c2) {
              const double x2 = sm.sa[c2];
              cnt += (x2 > x || (x2 == x && c2 > c)) ? 1 : 0;
            }
            sm.sb[lo + cnt] = x;
          }
          __syncthreads();
          // ---- cutoff (psis.py:135-141); R < 700: the log(DBL_MIN) floor cannot bind -------------
          const double xcut = sm.sb[M];
          int n = M;
          while (n > 0 && sm.sb[n - 1] == xcut) --n;  // ties at the cutoff leave the tail
          const double e_cut = exp(xcut);
          double acc_w = 0.0, acc_r = 0.0, acc_e = 0.0;
          bool smoothed = false;
          if (n > 4 && !(F.debug_skip & 8)) {
            __syncthreads();
            for (int j = lane; j < n; j += kWave) sm.sa[j] = exp(sm.sb[n - 1 - j]) - e_cut;  // psis.py:147
            __syncthreads();
            const double* y = sm.sa;
            // ---- 6. GPD fit (psis.py:163-208), lane j <-> grid point b_j -------------------------
            const int mest = 30 + isqrt_i(n);
            const double yq = y[((n + 2) >> 2) - 1];
            const double yn = y[n - 1];
            const bool act = lane < mest;
            double b = 1.0 - sqrt((double)mest / ((double)(lane + 1) - 0.5));  // psis.py:186
            b /= 3.0 * yq;                                                      // psis.py:187
            b += 1.0 / yn;                                                      // psis.py:188
            const double b_first = uniform_d(b);                        // lane 0: most negative
            const double b_last = uniform_d(__shfl(b, mest - 1));
            const double fbig = fma(-b_first, yn, 1.0), fsmall = fma(-b_last, yn, 1.0);
            const bool wide = (fbig < 0x1p100) && (fsmall > 0x1p-100);
            ProdAcc acc;
            acc.init();
            const double nb = -b;
            int i = 0;
            if (wide) {
              for (; i + 8 <= n; i += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc.mul(fma(nb, y[i + u], 1.0));
                acc.renorm();
              }
              for (; i < n; ++i) acc.mul(fma(nb, y[i], 1.0));
              acc.renorm();
            } else {
              for (; i < n; ++i) { acc.mul(fma(nb, y[i], 1.0)); acc.renorm(); }
            }
            const double kj = acc.log_value() / (double)n;                     // psis.py:190
            double ls = (double)n * (log(-(b / kj)) - kj - 1.0);               // psis.py:191
            const bool anynan = __ballot(act && (ls != ls)) != 0ull;
            const double lmax = wave_all<R_MAX>(act ? ls : -INF);
            double w = act ? exp(ls - lmax) : 0.0;                             // psis.py:192
            const double se = wave_all<R_SUM>(w);
            w = anynan ? qnan() : w / se;
            const bool keep = act && (w >= 10.0 * kEps);                       // psis.py:194-197
            const double sw = wave_all<R_SUM>(keep ? w : 0.0);
            const double b_post = wave_all<R_SUM>(keep ? b * (w / sw) : 0.0);  // psis.py:198,201
            double lp = 0.0;
            for (int ii = lane; ii < n; ii += kWave) lp += log1p(-b_post * y[ii]);  // psis.py:203
            const double k_post = wave_all<R_SUM>(lp) / (double)n;
            const double sigma = -k_post / b_post;                             // psis.py:205
            khat = ((double)n * k_post + 5.0) / ((double)n + 10.0);           // psis.py:206
            if (isfinite(khat)) {
              smoothed = true;
              for (int j = lane; j < n; j += kWave) {
                const double p = ((double)j + 0.5) / (double)n;                // psis.py:153
                double q;
                if (sigma <= 0.0) {
                  q = qnan();                                                  // psis.py:214-215
                } else {
                  const double l1 = log1p(-p);
                  q = (fabs(khat) < kEps) ? -l1 : expm1(-khat * l1) / khat;    // psis.py:218-221
                  q *= sigma;
                }
                double wj = q + e_cut;   // exp(log(q + e_cut)), psis.py:155
                if (wj > 1.0) wj = 1.0;  // psis.py:157
                const double ej = y[j] + e_cut;
                acc_w += wj;
                acc_r += wj / ej;
                acc_e += ej;
              }
            }
          }
          s1 = wave_all<R_SUM>(s1);
          s2 = wave_all<R_SUM>(s2);
          double total = s1;
          if (smoothed) {
            acc_w = wave_all<R_SUM>(acc_w);
            acc_r = wave_all<R_SUM>(acc_r);
            acc_e = wave_all<R_SUM>(acc_e);
            total = (s1 - acc_e) + acc_w;
          }
          const double L = log(total);                                          // psis.py:158
          const double A = (-m) - L;
          loo = smoothed ? A + log((double)(S - n) + acc_r) : A + log((double)S);
          lppd = log(s2) + ((-mn) - log((double)S));                          // loo.py:329-337
          if ((!(total > 1e-280) || !isfinite(loo) || !isfinite(lppd)) && !F.debug_skip) slow = true;
        }
      }
    }
    if (lane == 0) {
      if (slow) {
        const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
        F.slow_list[idx] = (unsigned)r;
      } else {
        if (P.diag) P.diag[r] = khat;
        if (P.loo_i) P.loo_i[r] = P.scale_value * loo;
        if (P.lppd_i) P.lppd_i[r] = lppd;
      }
    }
  }
}

}  // namespace pla
