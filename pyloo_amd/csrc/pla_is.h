// Streaming LOO pass for the two importance-sampling methods without a tail fit:
//   SIS  (sis.py:86-106)   lw = x - LSE(x),                         x = -ll - max(-ll)
//   TIS  (tis.py:91-120)   lw = x_t - LSE(x_t),  x_t = min(x, LSE(x) - log S + 0.5 log S)
// with the diagnostic ESS = 1 / sum_s exp(lw_s)^2 (sis.py:104-105, tis.py:118-119) and, fused as in
// pla_psis_loo, loo_i = LSE_s(lw_s + ll_s) (loo.py:289,319-324) and lppd_i (loo.py:329-337).
// In terms of three sums over the row (all of positive terms, nothing cancels):
//   A = sum e^x_t,  B = sum e^(2 x_t),  D = sum e^(x_t - x),  C = sum e^-x
//   ESS = A^2 / B,   loo_i = -max - log A + log D,   lppd_i = log C - max - log S
// (SIS: x_t = x, D = S).  One wavefront per observation, the row in its registers (S <= 4096, unit
// draw stride, 16-byte aligned rows): one HBM read; TIS looks through the registers once more for the (few) draws above
// the truncation point; the next row streams in behind the last pass.  Rows with non-finite entries or more than 690 nats of
// range go to the general kernel through the device list, as in pla_wave.h.
#pragma once

#include "pla_wave.h"

namespace pla {

// LW: the weights-returning flavour (compute_importance_weights(method="sis" | "tis"), base.py:146-166): the input is the log
// ratios themselves (raw = input), the outputs are lw (same dtype as the input) and the ESS; nothing streams in behind the
// sums -- the registers are stored as lw = min(x, cut) - log A first, each vector replaced by the next row's as it leaves.
template <typename T, int VEC, bool TIS, bool LW = false>
__global__ __launch_bounds__(kWave * kWavesPerBlock, 2) void is_wave_kernel(RowsParams P, FastParams F) {
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
  constexpr bool tis = TIS;
  const double log_S = log((double)S);
  const T* base = reinterpret_cast<const T*>(P.in);
  const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nw = (int64_t)gridDim.x * kWavesPerBlock;
  T v[kWaveSlots];
  if (w0 < P.n_obs) issue_row_loads<T, VEC>(v, base + PLA_ROW_OFFSET(P, w0), S);
  for (int64_t r = w0; r < P.n_obs; r += nw) {
    const int64_t rn = r + nw;
    const T* rp_next = rn < P.n_obs ? base + PLA_ROW_OFFSET(P, rn) : nullptr;
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(rp_next ? rp_next : base), 0, rp_next ? S * (int)sizeof(T) : 0, 0x00020000);
    const int nvec = __builtin_amdgcn_readfirstlane(P.n_draws) / VEC;  // (kept inside the loop: see pla_waic.h)
    const int qfull = nvec / kWave, qrem = nvec - qfull * kWave;
    // slots past the row: copies of the lane's first vector, whose contribution is subtracted afterwards
    pad_tail<T, VEC, NQ - 1, false>(v, qfull, qrem, (T)0);
    const double ncopy = (double)((NQ - qfull) - (lane < qrem ? 1 : 0));
    double first[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) first[e] = LW ? (double)v[e] : -(double)v[e];  // raw = -ll (weights mode: the input)
    // ---- statistics ------------------------------------------------------------------------------
    double mx, mn;
    {
      T cur = (T)(-pinf()), vmx = LW ? (T)pinf() : (T)(-pinf());
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        cur = vmax_nc<!LW>(v[i], cur);                                        // max raw
        vmx = LW ? vmin_nc(v[i], vmx) : vmax_nc<false>(v[i], vmx);            // min raw (LOO: as max ll)
      }
      mx = (double)cur;
      mn = LW ? (double)vmx : -(double)vmx;
    }
    double m, nmn;
    wave_all2<R_MAX>(mx, -mn, m, nmn);  // min = -max(-.)
    mn = -nmn;
    const double R = m - mn;
    bool slow = !(R < kWaveMaxRange);  // also inf / NaN through the maximum or the minimum
    double ess = 0.0, loo = 0.0, lppd = 0.0;
    // ---- pass 1: A = sum e^x, B = sum e^2x (SIS), C = sum e^-x ------------------------------------
    double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      double ep, en = 0.0;
      // x in [-R, 0], R < 690; a NaN draw poisons the sums -> general kernel.  (Weights mode has no use for e^-x.)
      if constexpr (LW) ep = exp_tab((double)v[i] - m, tab);
      else exp_pair((-(double)v[i]) - m, tab, ep, en);
      sa += ep;
      sb = fma(ep, ep, sb);
      sc += en;
      // pin the running sums: otherwise the scheduler starts all 64 independent exponentials at once and spills
      if (LW || (i & 1) == 1) asm volatile("" : "+v"(sa), "+v"(sb), "+v"(sc));  // (weights mode keeps the whole row live: one draw at a time)
      if constexpr (!tis && !LW)
        if ((i % VEC) == VEC - 1) issue_row_vector<T, VEC>(v, rs_next, i / VEC);
    }
    {
      double a0 = 0.0, b0 = 0.0, c0 = 0.0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double ep, en;
        exp_pair(first[e] - m, tab, ep, en);
        a0 += ep;
        b0 = fma(ep, ep, b0);
        c0 += en;
      }
      sa = fma(-ncopy, a0, sa);
      sb = fma(-ncopy, b0, sb);
      sc = fma(-ncopy, c0, sc);
    }
    double A, C, B = 0.0, D = (double)S;
    double cut_w = 0.0;  // (weights mode, TIS: the truncation point, for the output pass)
    wave_all2<R_SUM>(sa, sc, A, C);
    if constexpr (!tis) {
      B = wave_all<R_SUM>(sb);
    } else {
      // ---- pass 2 (TIS): truncate at cut = LSE(x) - log S + 0.5 log S (tis.py:107-110) ---------------------------
      // At most sqrt(S) draws can lie above the cut (their weights exceed A / sqrt(S) and sum to at most A), so the pass does
      // not evaluate the row again: one compare per slot finds them, and what truncation takes away from them is subtracted
      // from the sums of pass 1 --  A_t = A - sum_tr (e^x - e^cut),  B_t = B - sum_tr (e^2x - e^2cut),
      // D = S - sum_tr (1 - e^(cut - x))  -- with the SAME exponentials pass 1 added, so exactly those leave.  Nothing
      // cancels badly: A_t >= A / sqrt(S) and B_t >= A^2 / S >= B / S keep all but ~4 of the 16 digits.
      const double cut = (log_tab(A, lt) - log_S) + 0.5 * log_S;
      const double ecut = exp_tab(fmin(fmax(cut, -700.0), 700.0), tab);
      cut_w = cut;
      B = wave_all<R_SUM>(sb);
      double m2 = m;
      asm volatile("" : "+v"(m2));
      // v < thr  <=>  (-v) - m > cut, up to the rounding of the subtraction: thr is widened, the exact test follows inside
      const double thr0 = LW ? (m2 + cut) : -(m2 + cut);  // (weights mode: v > thr  <=>  v - m > cut)
      const double thr = thr0 + fmax(fabs(thr0), fabs(m2)) * 1e-12;
      const double thr_lw = thr0 - fmax(fabs(thr0), fabs(m2)) * 1e-12;
      double da = 0.0, db = 0.0, dd = 0.0;
      const auto term = [&](double x, double& a, double& b2, double& d) {
        double ep, en = 0.0;
        if constexpr (LW) ep = exp_tab(x, tab);
        else exp_pair(x, tab, ep, en);
        const bool tr = x > cut;
        a += tr ? ep - ecut : 0.0;
        b2 += tr ? fma(ep, ep, -ecut * ecut) : 0.0;
        if constexpr (!LW) d += tr ? fma(-ecut, en, 1.0) : 0.0;
      };
      if constexpr (LW) {
        // Weights mode keeps the row in the registers for the output pass, so there is no room for 64 conditional
        // exponentials next to it: one compare per slot collects the lane's candidates in a 64-bit mask, and the few draws it
        // names are read again from the row in memory (cache hits).  Slots past the row are skipped, not counted and corrected.
        // (Tried for the LOO pass as well: 6.3 instead of 5.7 ms -- there the conditional blocks ride under the streaming loads.)
        unsigned long long mask = 0ull;
#pragma unroll
        for (int i = 0; i < EPT; ++i) mask |= ((double)v[i] > thr_lw) ? (1ull << i) : 0ull;
        if (__ballot(mask != 0ull) != 0ull) {
          const T* rowp = base + PLA_ROW_OFFSET(P, r);
          while (mask) {
            const int i = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            const int idx = VEC * (lane + kWave * (i / VEC)) + i % VEC;
            if (idx < S) term((double)rowp[idx] - m2, da, db, dd);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          if (__ballot((double)v[i] < thr) != 0ull) term((-(double)v[i]) - m2, da, db, dd);
          if ((i % VEC) == VEC - 1) issue_row_vector<T, VEC>(v, rs_next, i / VEC);
        }
        // the padded slots hold copies of the lane's first vector
        double a0 = 0.0, b0 = 0.0, d0 = 0.0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) term(first[e] - m2, a0, b0, d0);
        da = fma(-ncopy, a0, da);
        db = fma(-ncopy, b0, db);
        dd = fma(-ncopy, d0, dd);
      }
      double sda, sdb;
      wave_all2<R_SUM>(da, db, sda, sdb);
      A -= sda;
      B -= sdb;
      D -= wave_all<R_SUM>(dd);
    }
    ess = div_fast(A * A, B);
    const double lg = log_tab(lane == 1 ? C : (lane == 2 ? D : A), lt);
    const double logA = lane_value(lg, 0), logC = lane_value(lg, 1), logD = lane_value(lg, 2);
    loo = ((-m) - logA) + logD;
    lppd = (logC - m) - log_S;
    if (!isfinite(ess) || !isfinite(logA) || (!LW && (!isfinite(loo) || !isfinite(lppd)))) slow = true;
    if constexpr (LW) {
      // lw = min(x, cut) - log A (sis.py:101-103, tis.py:112-116), from the registers; the next row's vectors take their place
      // (a declined row goes through the same code with an empty output range -- its stores are dropped -- so that the next
      // row's loads have ONE place of issue: two of them made the register allocator shuffle the row through scratch)
      T* orow = reinterpret_cast<T*>(P.lw_out) + r * (int64_t)S;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(orow, 0, slow ? 0 : S * (int)sizeof(T), 0x00020000);
      double m3 = m;  // laundered: otherwise the shifted values of pass 1 are kept alive for this pass and spill
      asm volatile("" : "+v"(m3));
      lw_store_chunk<T, VEC, true, TIS>(v, ro, rs_next, lane, qfull, m3, logA, cut_w);
    }
    if (lane == 0) {
      if (slow) {
        const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
        F.slow_list[idx] = (unsigned)r;
      } else {
        if (P.diag) P.diag[r] = ess;
        if constexpr (!LW) {
          if (P.loo_i) P.loo_i[r] = P.scale_value * loo;
          if (P.lppd_i) P.lppd_i[r] = lppd;
        }
      }
    }
  }
}

}  // namespace pla
