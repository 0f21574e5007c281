// WAIC kernels (reference: pyloo waic.py:109-160).  Per observation, from ONE read of its row of S
// log-likelihood draws:
//   lppd_i = logsumexp_s(ll) - log S              (waic.py:137-143, utils.py:305-359 with b_inv = S)
//   var_i  = population variance of ll over draws (waic.py:145, two-pass like np.var)
//   waic_i = scale * (lppd_i - var_i)             (waic.py:158)
// NaN -> -1e10 and +-inf -> +-1e10 are applied on load exactly as the reference front does before the
// arithmetic (waic.py:112-135); the number of replaced entries is counted so that the front can emit
// the reference's warnings.
//
// waic_wave_kernel: one wavefront per observation, the row in its registers (S <= 4096, unit draw stride,
// 16-byte aligned rows) -- one HBM read, everything else in registers; the next row streams into the
// registers the second pass has consumed (64 * 16/sizeof(T) <= S).  waic_rows_kernel: any shape / stride, one workgroup per
// observation, three passes over the (L2-resident) row.
#pragma once

#include "pla_wave.h"

namespace pla {

struct WaicParams {
  const void* in;
  int64_t n_obs;
  int n_draws;
  int64_t stride_obs, stride_draw;  // elements
  double scale_value;
  double* lppd_i;   // [n_obs] or null
  double* var_i;    // [n_obs] or null
  double* waic_i;   // [n_obs] or null
  unsigned long long* replaced;  // [1] device counter: NaN / inf entries replaced (may be null)
  const int64_t* row_index = nullptr;  // optional row selection, as in RowsParams
};

// waic.py:112-135
template <typename T>
__device__ __forceinline__ T waic_sanitize(T x, unsigned& nrep) {
  const bool nan = x != x;
  const bool inf = !nan && (x - x != (T)0);  // +-inf
  if (nan || inf) ++nrep;
  if (nan) return (T)-1e10;
  if (inf) return x > (T)0 ? (T)1e10 : (T)-1e10;
  return x;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kWave * kWavesPerBlock, 2) void waic_wave_kernel(WaicParams P) {
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  __shared__ __attribute__((aligned(16))) double lt[2 * kLogTabN];
  constexpr int EPT = kWaveSlots, NQ = EPT / VEC;
  const int tid = threadIdx.x;
  for (int j = tid; j < kTabN; j += kWave * kWavesPerBlock) exp_table_entry(tab, j);
  for (int j = tid; j < kLogTabN; j += kWave * kWavesPerBlock) log_table_entry(lt, j);
  __syncthreads();
  const int lane = wave_lane();
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int S = P.n_draws;
  const double inv_S = recip_fast((double)S), log_S = log((double)S);
  const T* base = reinterpret_cast<const T*>(P.in);
  const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nw = (int64_t)gridDim.x * kWavesPerBlock;
  T v[kWaveSlots];
  if (w0 < P.n_obs) issue_row_loads<T, VEC>(v, base + PLA_ROW_OFFSET(P, w0), S);
  unsigned nrep = 0;
  for (int64_t r = w0; r < P.n_obs; r += nw) {
    const int64_t rn = r + nw;
    const T* rp_next = rn < P.n_obs ? base + PLA_ROW_OFFSET(P, rn) : nullptr;
    // (read through readfirstlane inside the loop: otherwise the 31 per-vector masks of pad_tail are
    // hoisted out of the row loop and spill)
    const int nvec = __builtin_amdgcn_readfirstlane(P.n_draws) / VEC;
    const int qfull = nvec / kWave, qrem = nvec - qfull * kWave;
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(rp_next ? rp_next : base), 0, rp_next ? S * (int)sizeof(T) : 0, 0x00020000);
    // Slots past the row become copies of the lane's first vector: max / sums need no per-slot predicate,
    // and the exactly known contribution of the copies is subtracted afterwards.
    pad_tail<T, VEC, NQ - 1, false>(v, qfull, qrem, (T)0);
    const double ncopy = (double)((NQ - qfull) - (lane < qrem ? 1 : 0));  // copies of the first vector in this lane
    double first[VEC];
    double mx, sum;
    const auto pass1 = [&]() {
#pragma unroll
      for (int e = 0; e < VEC; ++e) first[e] = (double)v[e];
      mx = -pinf();
      sum = 0.0;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        mx = fmax(mx, (double)v[i]);
        sum += (double)v[i];
      }
      double s0 = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) s0 += first[e];
      sum = wave_all<R_SUM>(fma(-ncopy, s0, sum));
    };
    // ---- pass 1: max, sum.  A NaN or an infinity anywhere in the row makes the sum non-finite: only
    // then are the replacements of waic.py:112-135 applied (and counted) and the pass repeated.
    pass1();
    if (!isfinite(sum)) {
      // every slot holds a draw of the row (the tail slots: copies of the first vector, counted back out)
      unsigned rep_all = 0, rep_first = 0;
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        unsigned rep = 0;
        v[i] = waic_sanitize(v[i], rep);
        rep_all += rep;
        if (i < VEC) rep_first += rep;
      }
      nrep += rep_all - (unsigned)ncopy * rep_first;
      pass1();
    }
    const double m = wave_all<R_MAX>(mx);
    const double mean = sum * inv_S;
    // ---- pass 2: sum exp(x - m) and sum (x - mean)^2; the next row streams in behind it ----------
    double se = 0.0, sq = 0.0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const double x = (double)v[i];
      se += exp_tab(fmax(x - m, -700.0), tab);
      const double d = x - mean;
      sq = fma(d, d, sq);
      if ((i % VEC) == VEC - 1) issue_row_vector<T, VEC>(v, rs_next, i / VEC);
    }
    {
      double e0 = 0.0, q0 = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        e0 += exp_tab(fmax(first[e] - m, -700.0), tab);
        const double d = first[e] - mean;
        q0 = fma(d, d, q0);
      }
      se = fma(-ncopy, e0, se);
      sq = fma(-ncopy, q0, sq);
    }
    se = wave_all<R_SUM>(se);
    const double var = wave_all<R_SUM>(sq) * inv_S;
    const double lppd = (log_tab(se, lt) + m) - log_S;  // utils.py:352-357
    if (lane == 0) {
      if (P.lppd_i) P.lppd_i[r] = lppd;
      if (P.var_i) P.var_i[r] = var;
      if (P.waic_i) P.waic_i[r] = P.scale_value * (lppd - var);
    }
  }
  if (P.replaced) {
    const unsigned tot = (unsigned)wave_all<R_SUM>((double)nrep);
    if (lane == 0 && tot) atomicAdd(P.replaced, (unsigned long long)tot);
  }
}

template <typename T, int BLOCK>
__global__ __launch_bounds__(BLOCK) void waic_rows_kernel(WaicParams P) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  const int S = P.n_draws;
  unsigned nrep = 0;
  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const T* rp = reinterpret_cast<const T*>(P.in) + PLA_ROW_OFFSET(P, r);
    double mx = -pinf(), sum = 0.0;
    for (int s = tid; s < S; s += BLOCK) {
      const double x = (double)waic_sanitize(rp[(int64_t)s * P.stride_draw], nrep);
      mx = fmax(mx, x);
      sum += x;
    }
    const double m = block_reduce<OpMax, BLOCK>(mx, red);
    const double mean = block_reduce<OpSum, BLOCK>(sum, red) / (double)S;
    double se = 0.0, sq = 0.0;
    unsigned dummy = 0;
    for (int s = tid; s < S; s += BLOCK) {
      const double x = (double)waic_sanitize(rp[(int64_t)s * P.stride_draw], dummy);
      se += exp(x - m);
      sq += (x - mean) * (x - mean);
    }
    se = block_reduce<OpSum, BLOCK>(se, red);
    const double var = block_reduce<OpSum, BLOCK>(sq, red) / (double)S;
    const double lppd = (log(se) + m) - log((double)S);
    if (tid == 0) {
      if (P.lppd_i) P.lppd_i[r] = lppd;
      if (P.var_i) P.var_i[r] = var;
      if (P.waic_i) P.waic_i[r] = P.scale_value * (lppd - var);
    }
  }
  if (P.replaced) {
    const double tot = block_reduce<OpSum, BLOCK>((double)nrep, red);
    if (tid == 0 && tot > 0.0) atomicAdd(P.replaced, (unsigned long long)tot);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Observations-fastest matrices (ArviZ's layout behind pyloo's stacked view, waic.py:104-107): ONE LANE PER OBSERVATION,
// like pla_col.h -- a wave reads draw s of 64 neighbouring observations as one contiguous piece of the matrix and every lane
// streams down its own observation, in one pass:
//   * log-sum-exp with a running maximum: a draw above it rescales the sum (a second exponential, taken by the wave only
//     when one of its lanes meets a new maximum -- ~log S times per lane);
//   * the variance by batches of U draws: two-pass inside the batch (its draws sit in registers), merged into the running
//     (n, mean, M2) with the pairwise update of Chan, Golub & LeVeque -- as accurate as np.var's two passes over the row.
// No transposing pass through HBM (19.7 ms for C3 in round 1): the matrix is read once.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef PLA_WAIC_COL_BATCHES
#define PLA_WAIC_COL_BATCHES 3
#endif
template <typename T>
__global__ __launch_bounds__(256) void waic_col_kernel(WaicParams P, int64_t ld) {
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  __shared__ __attribute__((aligned(16))) double lt[2 * kLogTabN];
  for (int j = threadIdx.x; j < kTabN; j += 256) exp_table_entry(tab, j);
  for (int j = threadIdx.x; j < kLogTabN; j += 256) log_table_entry(lt, j);
  __syncthreads();
  const int S = P.n_draws;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < P.n_obs;
  const T* col = reinterpret_cast<const T*>(P.in) + (live ? i : P.n_obs - 1);
  constexpr int U = 8;
  unsigned nrep = 0;
  double m = -pinf(), se = 0.0;      // running maximum, sum of exp(x - m)
  double mean = 0.0, m2 = 0.0;       // Chan: mean and sum of squared deviations of the draws seen so far
  const auto batch = [&](const double (&x)[U], const int n_before, const int nb) {
    // ---- log-sum-exp ----
    double bm = x[0];
#pragma unroll
    for (int u = 1; u < U; ++u) bm = (u < nb) ? fmax(bm, x[u]) : bm;
    if (__ballot(bm > m) != 0ull) {  // some lane has a new maximum: rescale what it has summed so far
      const double mnew = fmax(m, bm);
      se *= exp_tab(fmax(m - mnew, -700.0), tab);  // (exp(-inf) on the first batch: se is 0 anyway)
      m = mnew;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (u < nb) se += exp_tab(fmax(x[u] - m, -700.0), tab);
    // ---- variance ----
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) sum += (u < nb) ? x[u] : 0.0;
    const double mb = sum / (double)nb;
    double qb = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double d = x[u] - mb;
      qb = (u < nb) ? fma(d, d, qb) : qb;
    }
    const double tot = (double)(n_before + nb);
    const double delta = mb - mean;
    m2 += qb + delta * delta * ((double)n_before * (double)nb / tot);
    mean += delta * ((double)nb / tot);
  };
  // NB batches of U draws per lane in flight: the loads of batch b + NB are issued into the registers of batch b as soon as
  // that batch has been converted, before its arithmetic (one batch, loaded and then computed on, left the kernel at 4.5 TB/s
  // on occupancy alone; tools/microbench/col_stream.hip: this access shape streams at 6.5)
  constexpr int NB = PLA_WAIC_COL_BATCHES;
  T ring[NB][U];
  const auto fetch = [&](T (&dst)[U], const int s0) {  // (draws past the row: its last draw again, never looked at)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int d = s0 + u < S ? s0 + u : S - 1;
      dst[u] = __builtin_nontemporal_load(col + (int64_t)d * ld);
    }
  };
#pragma unroll
  for (int b = 0; b < NB; ++b) fetch(ring[b], b * U);
  const auto take = [&](const int b, const int s0, const int nb) {
    double x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned rep = 0;
      x[u] = (double)waic_sanitize(ring[b][u], rep);  // waic.py:112-135
      nrep += (u < nb && live) ? rep : 0u;
    }
    fetch(ring[b], s0 + NB * U);
    batch(x, s0, nb);
  };
  int s = 0;
#pragma unroll 1
  for (; s + NB * U <= S; s += NB * U) {  // whole rounds: no control flow between the batches (the load counters stay exact)
#pragma unroll
    for (int b = 0; b < NB; ++b) take(b, s + b * U, U);
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {  // what is left of the row (already in the ring)
    const int s0 = s + b * U;
    const int nb = S - s0 < U ? S - s0 : U;  // (wave-uniform)
    if (nb > 0) take(b, s0, nb);
  }
  if (live) {
    const double var = m2 / (double)S;
    const double lppd = (log_tab(se, lt) + m) - log((double)S);  // utils.py:352-357
    if (P.lppd_i) P.lppd_i[i] = lppd;
    if (P.var_i) P.var_i[i] = var;
    if (P.waic_i) P.waic_i[i] = P.scale_value * (lppd - var);
  }
  if (P.replaced) {
    __shared__ double red[4];
    const double tot = block_reduce<OpSum, 256>((double)nrep, red);
    if (threadIdx.x == 0 && tot > 0.0) atomicAdd(P.replaced, (unsigned long long)tot);
  }
}

}  // namespace pla
