// Output pass of the weights-returning split pass for rows longer than the registers (psislw / compute_importance_weights on
// S > 4096 draws; base.py:160-166 output, psis.py:150-158).
//
// The fused weights kernel of rounds 2-3 (pla_chunked.h, LW without SPLIT) keeps ONE six-wave workgroup per CU (the fit's
// tables and the candidates' draw indices in LDS) and walks a row three times inside one wave: 0.24 of the HBM peak.  Here the
// three stages are three kernels, each at the occupancy its own state allows:
//   1. wave_loo_chunked_kernel<T, VEC, CapsMid, SPLIT, LW>: statistics, sweep, selection of the tail -- the LOO pass's kernel
//      with the sign convention of weights mode, eight waves per CU; hands over the tail's shifted log ratios x and
//      (max, min, sum e^x, -, cutoff, tail length);
//   2. fit_rows_kernel (pla_fit.h) with FitParams::lw_mode: sorts the tail, fits, smooths, and leaves per observation the SORTED
//      tail x (descending) in place of the hand-over, log(sum of all weights), the number of draws to patch and the three scalars
//      the smoothed weight of a rank is evaluated from (k-hat, sigma / k-hat, e_cut - sigma / k-hat);
//   3. this kernel: one wave per observation streams the row once more -- lw = (raw - max raw) - log sum for every draw
//      (psis.py:134,158) -- and collects the draws above the cutoff (psis.py:139-141: exactly the tail) with their positions;
//      each of them then finds its descending rank in the sorted tail by bisection (equal draws take consecutive ranks through
//      a counter per rank: which of them gets which quantile is arbitrary in the reference too, its argsort is unstable,
//      psis.py:146) and its position is overwritten with log(smoothed weight of that rank) - log sum (psis.py:155-158); the
//      weight of a rank is the fit kernel's own expression evaluated here (the weights by rank read from memory -- a dependent
//      load behind the bisection -- cost 4 % of the pass).
// No draw index travels through the selection, the sort or the fit.  Observations whose scalar [5] is negative were put on
// the list for the general kernel (by the selection or by the fit): that kernel writes their rows, this one skips them.
#pragma once

#include "pla_wave.h"

namespace pla {

constexpr int kLwoTail = 448;   // longest tail the split pass hands over (CapsMid::kMaxTail)
constexpr int kLwoWaves = 4;
constexpr int kLwoBatch = 8;   // 16-byte vectors in flight per lane

struct LwOutParams {
  const void* in;       // (n_obs, n_draws), unit draw stride, 16-byte aligned rows
  void* out;            // (n_obs, n_draws) contiguous, input dtype
  int64_t n_obs;
  int n_draws;
  int64_t stride_obs;   // elements
  const double* ws_y;   // [n_obs][ws_stride] sorted tail x, descending
  const double* ws_s;   // [n_obs][ws_sstride]: max raw, -, log sum, k-hat, cutoff, tail draws to patch (< 0: not this kernel's row), sigma / k-hat, e_cut - sigma / k-hat
  int ws_stride;
  int ws_sstride;
  const double* l1_table;  // [tail_count] log1p(-(j + 0.5) / M) (host-computed, as the fit kernel reads it)
  int tail_count;
};


struct LwoSmem {                    // per wave: 11.6 KB -> three four-wave workgroups per CU
  double xs[kLwoTail];              // sorted tail
  double lx[kLwoTail + kWave];      // collected tail draws: x ...
  unsigned li[kLwoTail + kWave];    // ... and position in the row
  unsigned cnt[kLwoTail / 2];       // equal draws: how many took this rank already (two 16-bit counters to a word)
};

template <typename T>
__global__ __launch_bounds__(kWave * kLwoWaves, 3) void lw_output_kernel(LwOutParams P) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef int v4i __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) double lt[2 * kLogTabN];
  __shared__ __attribute__((aligned(16))) LwoSmem smem[kLwoWaves];
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  __shared__ __attribute__((aligned(16))) double l1s[kLwoTail];
  const int tid = threadIdx.x;
  for (int j = tid; j < kLogTabN; j += kWave * kLwoWaves) log_table_entry(lt, j);
  for (int j = tid; j < kTabN; j += kWave * kLwoWaves) exp_table_entry(tab, j);
  const int M = P.tail_count;
  for (int j = tid; j < M; j += kWave * kLwoWaves) l1s[j] = P.l1_table[j];
  __syncthreads();
  const int lane = tid & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  LwoSmem& sm = smem[wv];
  const int S = __builtin_amdgcn_readfirstlane(P.n_draws);
  const int nvec = (S + VEC - 1) / VEC;                  // 16-byte vectors of a row (the last one may be partial)
  const int rounds = (nvec + kWave - 1) / kWave;         // vectors per lane
  const int64_t w0 = (int64_t)blockIdx.x * kLwoWaves + wv, nw = (int64_t)gridDim.x * kLwoWaves;
  const T* base = reinterpret_cast<const T*>(P.in);
  T* obase = reinterpret_cast<T*>(P.out);
  for (int64_t r = w0; r < P.n_obs; r += nw) {
    const double* sc = P.ws_s + r * P.ws_sstride;
    const double nd = uniform_d(sc[5]);
    if (nd < 0.0) continue;  // (wave-uniform: the general kernel writes this row)
    const int n = (int)nd;
    const double m = uniform_d(sc[0]), L = uniform_d(sc[2]), xcut = uniform_d(sc[4]);
    const double khat = uniform_d(sc[3]), coef_s = uniform_d(sc[6]), off = uniform_d(sc[7]);
    const double rn = n > 0 ? 1.0 / (double)n : 0.0;
    wave_sync();  // (the row before is done with the lists)
    if (n > 0) {
      const double* ys = P.ws_y + r * (int64_t)P.ws_stride;
      for (int j = lane; j < n; j += kWave) sm.xs[j] = ys[j];
      for (int j = lane; j < (n + 1) / 2; j += kWave) sm.cnt[j] = 0u;
    }
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base + r * P.stride_obs), 0, S * (int)sizeof(T), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(obase + r * (int64_t)S, 0, S * (int)sizeof(T), 0x00020000);
    unsigned ntail = 0;  // wave-uniform
    // two batches of vectors per lane take turns: the loads of one are in flight while the other is turned into weights
    const auto fetch = [&](v4i (&t)[kLwoBatch], const int q0) {
#pragma unroll
      for (int b = 0; b < kLwoBatch; ++b)  // (past the end of the row: zeros, and the stores of such vectors are dropped)
        t[b] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16, (q0 + b) * (kWave * 16), PLA_LOAD_AUX);
    };
    // The tail draws a batch has collected get their smoothed weights at once, right behind the batch's own stores: the lines
    // are then still in the L2 (a patch written after the WHOLE row finds them evicted -- 425 partial-line writes per row to
    // memory, 0.8 of 2.8 ms).  Same wave, same addresses, program order: the memory pipeline keeps the two stores in that order.
    T* const orow = obase + r * (int64_t)S;
    const auto patch = [&](const unsigned nt) {
      wave_sync();
      constexpr int U = 2;
      for (unsigned c0 = lane; c0 < nt; c0 += U * kWave) {
        double x[U];
        int lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned c = c0 + u * kWave;
          x[u] = sm.lx[c < nt ? c : c0];
          lo[u] = 0;
          hi[u] = n;  // descending: the number of sorted values above x
        }
#pragma unroll 1
        for (int it = 0; it < 9; ++it) {  // 2^9 > 448
          double xm[U];
#pragma unroll
          for (int u = 0; u < U; ++u) xm[u] = sm.xs[(lo[u] + hi[u]) >> 1];  // (lo = hi = n = 448 reads one entry past xs: still this wave's scratch, not used)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int mid = (lo[u] + hi[u]) >> 1;
            const bool open = lo[u] < hi[u], above = open & (xm[u] > x[u]);
            lo[u] = above ? mid + 1 : lo[u];
            hi[u] = (open & !above) ? mid : hi[u];
          }
        }
        unsigned slot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = c0 + u * kWave < nt && lo[u] < n;
          const unsigned sh = ((unsigned)lo[u] & 1u) << 4;
          slot[u] = live ? (unsigned)lo[u] + ((atomicAdd(&sm.cnt[lo[u] >> 1], 1u << sh) >> sh) & 0xffffu) : (unsigned)n;
        }
        double w[U];
        // w_j = min(sigma / k (e^(-k log1p(-p_j)) - 1) + e_cut, 1) with p_j = (j + 0.5) / n, j the ASCENDING index of the rank
        // (psis.py:153-157,218-221): the fit kernel's own expression (pla_fit.h, smooth()), from the scalars it left
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = n - 1 - (int)(slot[u] < (unsigned)n ? slot[u] : 0u);
          const double l1 = (n == M) ? l1s[j] : log_fast(1.0 - ((double)j + 0.5) * rn);
          const double ez = exp_tab(fmin(-khat * l1, 700.0), tab);
          w[u] = fmin(fma(ez, coef_s, off), 1.0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned c = c0 + u * kWave;
          const double v = log_tab(w[u], lt) - L;                                   // psis.py:155-158
          if (slot[u] < (unsigned)n) orow[sm.li[c < nt ? c : c0]] = (T)v;
        }
      }
      wave_sync();  // (the list is free for the next batch)
    };
    const auto emit = [&](const v4i (&t)[kLwoBatch], const int q0) {
#pragma unroll
      for (int b = 0; b < kLwoBatch; ++b) {
        const int first = ((q0 + b) * kWave + lane) * VEC;  // position of this vector's first draw
        double x[VEC];
        if constexpr (VEC == 2) {
          x[0] = __hiloint2double(t[b][1], t[b][0]) - m;
          x[1] = __hiloint2double(t[b][3], t[b][2]) - m;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = (double)__int_as_float(t[b][e]) - m;  // psis.py:134
        }
        v4i o;
        if constexpr (VEC == 2) {
          const double a0 = x[0] - L, a1 = x[1] - L;                                // psis.py:158
          o[0] = __double2loint(a0); o[1] = __double2hiint(a0);
          o[2] = __double2loint(a1); o[3] = __double2hiint(a1);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = __float_as_int((float)(x[e] - L));
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, ro, lane * 16, (q0 + b) * (kWave * 16), 0);
        asm volatile("s_nop 0" : : "v"(o));  // (gfx9 hazard of > 8-byte buffer stores with a scalar offset: see lw_store_chunk)
        if (n > 0) {
          // the tail draws of this vector, appended to the wave's list: straight-line code (a draw that is no tail draw writes to
          // its lane's dump entry behind the list) -- with a branch per element this collection cost 0.8 of the kernel's 2.8 ms
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const bool tail = (x[e] > xcut) & (first + e < S);                       // psis.py:139-141
            const unsigned long long tm = __ballot(tail);
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(tm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tm, 0u));
            unsigned at = ntail + rank;
            at = at < (unsigned)kLwoTail ? at : (unsigned)(kLwoTail - 1);            // (cannot happen: the tail holds <= 448 draws)
            at = tail ? at : (unsigned)(kLwoTail + lane);
            sm.lx[at] = x[e];
            sm.li[at] = (unsigned)(first + e);
            ntail += (unsigned)__popcll(tm);
          }
        }
      }
#ifndef PLA_LWO_ABLATE
#define PLA_LWO_ABLATE 0  // timing experiments: 2 = the tail draws are collected but not patched
#endif
      if (ntail != 0u) {  // (wave-uniform)
        if (!(PLA_LWO_ABLATE & 2)) patch(ntail < (unsigned)kLwoTail ? ntail : (unsigned)kLwoTail);
        ntail = 0u;
      }
    };
    {
      v4i ta[kLwoBatch], tb[kLwoBatch];
      fetch(ta, 0);
      for (int q0 = 0; q0 < rounds; q0 += 2 * kLwoBatch) {
        fetch(tb, q0 + kLwoBatch);
        emit(ta, q0);
        fetch(ta, q0 + 2 * kLwoBatch);
        emit(tb, q0 + kLwoBatch);
      }
    }
  }
}

}  // namespace pla
