// PSIS-LOO for log-likelihood matrices with the OBSERVATIONS fastest -- the layout ArviZ keeps ((chain, draw, *obs) in
// memory; pyloo's stacked `(*obs, __sample__)` view of it, loo.py:189, is what `pl.loo(idata)` hands to the hot path).
//
// The row kernels (pla_wave.h) give every observation a wavefront and want its draws contiguous; for this layout that cost
// a transposing pass through HBM first (read + write + read: 21.5 ms for C3 against 7.3 ms, round 1).  Here the roles of
// lanes and slots are swapped instead: ONE LANE PER OBSERVATION.  A wave owns 64 neighbouring observations, so the draw s of
// all of them is one contiguous 512-byte (f64) piece of the matrix -- perfectly coalesced as it lies -- and every lane
// streams down its own observation:
//
//   col_sweep_kernel    one pass over the matrix.  Per observation (lane), draw by draw: max / min of raw = -ll, the two sums
//                       of e^x', e^-x' about a PROVISIONAL shift m' (as in pla_chunked.h; the table-driven exponentials of
//                       the wave kernel, same 19 VALU operations per draw -- but here they are the whole cost: no per-row
//                       statistics pass, no cross-lane reductions, all 64 lanes busy on every instruction), and the draws
//                       at or above a speculative threshold are appended to the observation's candidate list in a workspace.
//                       m' and the threshold come from a pre-pass over 512 draws spread evenly over the row (every chain of
//                       a chain-major stack contributes): the kq-th smallest of 64 group maxima, found by a per-lane bisection
//                       over 64 registers.
//   col_select_kernel   one wavefront per observation again, but on ~420 candidates instead of 4000 draws: true shift,
//                       x = raw - m with the reference's single rounding (psis.py:134), then the selection of the split
//                       pass (wave_select_split: histogram, scan, boundary bin) and the hand-over to fit_rows_kernel.
//
// HBM traffic: the matrix once + the 512 sampled draws a second time (+12.5 %) + the candidate lists (written and read:
// ~2 x 3.4 KB per observation at S = 4000).  Rows the shortcuts cannot take (non-finite draws, > 690 nats of range, a
// threshold miss) go to the general kernel through the same device list as everywhere else.
#pragma once

#include "pla_wave.h"

namespace pla {

constexpr int kColSample = 512;   // draws in the pre-pass: 64 groups of 8
constexpr int kColCap = 1024;     // candidate list capacity per observation (doubles)

struct ColParams {
  const void* in;      // element (observation i, draw s) at in[s * ld + i]
  int64_t n_obs;       // observations of this launch
  int n_draws;
  int64_t ld;          // elements between consecutive draws
  int kq;              // the threshold has kq of the 64 group maxima below it
  double* cand;        // [n_obs][kColCap] raw values at or above the threshold
  double* scal;        // [n_obs][8]: m', max raw, min raw, sum e^x', sum e^-x', number of candidates (uncapped), -, -
};

template <typename T>
__global__ __launch_bounds__(256) void col_sweep_kernel(ColParams P) {
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  for (int j = threadIdx.x; j < kTabN; j += 256) exp_table_entry(tab, j);
  __syncthreads();
  const int S = P.n_draws;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < P.n_obs;
  const T* col = reinterpret_cast<const T*>(P.in) + (live ? i : P.n_obs - 1);  // (idle lanes re-read the last observation)
  const double INF = pinf();

  // ---- pre-pass: 64 group maxima over 512 draws spread over the row; group g holds the samples g, g + 64, ... ----------
  float gm[64];
  double mp = -INF;  // provisional shift: the largest sampled raw value
#pragma unroll
  for (int g = 0; g < 64; ++g) gm[g] = -__builtin_inff();
#pragma unroll 1
  for (int k = 0; k < kColSample / 64; ++k) {
#pragma unroll
    for (int g = 0; g < 64; ++g) {
      const int s = (int)(((int64_t)(k * 64 + g) * S) / kColSample);
      const double raw = -(double)col[(int64_t)s * P.ld];
      mp = fmax(mp, raw);
      // rounded up: the threshold may only err towards FEWER candidates by what one float ulp is worth
      gm[g] = fmaxf(gm[g], __double2float_ru(raw));
    }
  }
  float lo = gm[0], hi = gm[0];
#pragma unroll
  for (int g = 1; g < 64; ++g) {
    lo = fminf(lo, gm[g]);
    hi = fmaxf(hi, gm[g]);
  }
#pragma unroll 1
  for (int it = 0; it < 12; ++it) {  // a value with >= kq group maxima below it (per lane: no cross-lane traffic)
    const float mid = 0.5f * (lo + hi);
    int below = 0;
#pragma unroll
    for (int g = 0; g < 64; ++g) below += (gm[g] < mid) ? 1 : 0;
    if (below >= P.kq) hi = mid;
    else lo = mid;
  }
  const double t_raw = (double)hi;

  // ---- the pass: every draw of the observation, 8 loads in flight per lane -----------------------------------------
  double mx = -INF, mn = INF, s1 = 0.0, s2 = 0.0;
  int cnt = 0;
  double* list = P.cand + (live ? i : 0) * (int64_t)kColCap;
  const char* tabc = reinterpret_cast<const char*>(tab);
  constexpr int U = 8;
  int c4096 = 4096, cm4096 = -4096, four = 4;
  asm volatile("" : "+s"(c4096), "+s"(cm4096));
  asm volatile("" : "+v"(four));
  const auto one = [&](double raw) {
    mx = fmax(mx, raw);
    mn = fmin(mn, raw);
    const double x = raw - mp;                       // psis.py:134 about the provisional shift
    const double t = fma(x, kC256, kMagic);
    const int k = __double2loint(t);                 // round(x * 256 / ln 2)
    const int4 tt = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));  // 16 * (k & 255)
    const double rr = fma(t - kMagic, -kLn2_256, x);
    const double r2 = rr * rr;
    const double E = fma(r2, 0.5, 1.0);              // cosh rr to 1.5e-13 (as in the wave kernel's sweep)
    const double O = fma(1.66666666666666666667e-01, r2, 1.0);
    s1 = fma(__hiloint2double(mad_i24(k, c4096, tt.y), tt.x), fma(rr, O, E), s1);
    s2 = fma(__hiloint2double(mad_i24(k, cm4096, tt.w), tt.z), fma(-rr, O, E), s2);
    if (raw >= t_raw) {
      if (cnt < kColCap && live) list[cnt] = raw;
      ++cnt;
    }
  };
  int s = 0;
#pragma unroll 1
  for (; s + U <= S; s += U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(col + (int64_t)(s + u) * P.ld);
#pragma unroll
    for (int u = 0; u < U; ++u) one(-(double)v[u]);
  }
  for (; s < S; ++s) one(-(double)col[(int64_t)s * P.ld]);
  if (live) {
    double* o = P.scal + i * 8;
    o[0] = mp; o[1] = mx; o[2] = mn; o[3] = s1; o[4] = s2; o[5] = (double)cnt;
  }
}

// One wavefront per observation: the candidate list -> LDS, then the split pass's selection.  Workgroups of 4 independent
// waves sharing the exponential table.
template <class CAP>
__global__ __launch_bounds__(kWave * 4) void col_select_kernel(ColParams P, FastParams F, int tail_count) {
  using SM = WaveSmemT<CAP>;
  struct TB { double tab[2 * kTabN]; };
  __shared__ __attribute__((aligned(16))) SM scratch[4];
  __shared__ __attribute__((aligned(16))) TB tb;
  for (int j = threadIdx.x; j < kTabN; j += kWave * 4) exp_table_entry(tb.tab, j);
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int lane = wave_lane();
  SM& sm = scratch[wv];
  const int M = tail_count;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < P.n_obs; r += (int64_t)gridDim.x * 4) {
    const double* sc = P.scal + r * 8;
    const double mp = uniform_d(sc[0]), m = uniform_d(sc[1]), mn = uniform_d(sc[2]);
    const double s1p = uniform_d(sc[3]), s2p = uniform_d(sc[4]);
    const int ncand = (int)uniform_d(sc[5]);
    const double R = m - mn, delta = m - mp;  // delta >= 0: the sample's maximum against the row's
    bool slow = !(R < kWaveMaxRange) || ncand < M + 1 || ncand > CAP::kCand || ncand > kColCap || !(fabs(s1p) < pinf()) ||
                !(fabs(s2p) < pinf());
    if (!slow) {
      wave_sync();  // the previous observation is done with the scratch
      {
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < kWaveBins / (4 * kWave); ++j) *reinterpret_cast<uint4*>(&sm.hist[4 * (lane + kWave * j)]) = z4;
      }
      const double* list = P.cand + r * (int64_t)kColCap;
      double xmin = 0.0;
      for (int c = lane; c < ncand; c += kWave) {
        const double x = list[c] - m;  // psis.py:134, one rounding, like the reference
        sm.cand[c] = x;
        xmin = fmin(xmin, x);
      }
      xmin = wave_all<R_MIN>(xmin);
      double magic = kMagic, c256 = kC256;
      const int k1 = __double2loint(fma(xmin, c256, magic));  // histogram origin: the smallest candidate's key
      const int span = -k1;
      const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
      // the sums about the true shift: e^x = e^x' e^-(m - m'), e^-x = e^-x' e^(m - m')   (R < 690 keeps both finite)
      const double s1 = lane == 0 ? s1p * exp_tab(-delta, tb.tab) : 0.0;
      const double s2 = lane == 0 ? s2p * exp_tab(delta, tb.tab) : 0.0;
      wave_sync();
      wave_select_split<SM, TB, (CAP::kMaxTail + 63) / 64>(F, sm, tb, r, lane, M, m, mn, s1, s2, (unsigned)ncand, k1, sh, magic, c256,
                                                           slow);
    }
    if (slow && lane == 0) {
      const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
      F.slow_list[idx] = (unsigned)r;
      F.ws_s[r * 8 + 5] = -1.0;  // tail length -1: on the list, nothing for the fit kernel
    }
  }
}

}  // namespace pla
