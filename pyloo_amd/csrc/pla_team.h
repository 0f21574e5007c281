// Team kernel: TWO wavefronts (one 128-thread workgroup) per observation.
//
// Same algorithm as the wave kernel (pla_wave.h), but each lane keeps only 32 of the row's draws
// (64 VGPRs for fp64), so a wave needs <= 128 registers and FOUR waves fit on every SIMD (16 per
// CU, 8 rows in flight per CU): twice the latency hiding of the one-wave-per-row layout, and
// every per-row phase is worked on by two waves in parallel.
//
// The two waves meet ~10 times per row at a raw s_barrier (never __syncthreads(): its fence would
// drain the next row's loads, which are issued right after the sweep and stay in flight through
// the selection / fit / smoothing of the current row).  What crosses waves goes through LDS:
//   wave maxima / minima / thresholds, each wave's own candidate list segment, the shared
//   histogram + scatter, the halves of the log-product of every GPD grid point, partial sums.
#pragma once

#include "pla_wave.h"

namespace pla {

constexpr int kTeamBlock = 128;
constexpr int kTeamSlots = 32;      // register slots per lane -> S <= 4096
#ifndef PLA_TEAM_SEG
#define PLA_TEAM_SEG 384
#endif
constexpr int kTeamSeg = PLA_TEAM_SEG;       // candidate capacity of each wave's list segment

// lgkmcnt(0) + s_barrier: LDS traffic of both waves is complete and visible; vector memory (the
// prefetched row) is left in flight.  The "memory" clobber keeps the compiler from moving LDS
// accesses across it.
__device__ __forceinline__ void team_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct TeamSmem {
  unsigned hist[kWaveBins];
  unsigned short start[kWaveBins];
  double cand[2 * kTeamSeg + kTeamBlock];  // two list segments + one dump slot per thread; later: sorted
                                           // candidates / pair sums and products, and the fit partials
  double sa[kWaveCap];                     // binned candidates, later y ascending
  double tab[2 * kTabN];
  double l1[kWaveMaxTail + 6];
  double bg[kWave];
  double xw[2][8];                         // per-wave scalars exchanged at the barriers
};
// per-wave (mantissa, exponent, correction) of every grid point: cand[384 + (w*64 + lane)*3 + c]
// (the list segments are dead by then; the sorted candidates / pair products use cand[0, 320))

template <typename T, int VEC>
__device__ __forceinline__ void issue_team_loads(T (&v)[kTeamSlots], const T* rp, int S) {
  constexpr int NQ = kTeamSlots / VEC;
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(rp), 0, S * (int)sizeof(T), 0x00020000);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, q * (kTeamBlock * 16), 2 /* nt */);
    if constexpr (VEC == 2) {
      v[2 * q] = (T)__hiloint2double(t[1], t[0]);
      v[2 * q + 1] = (T)__hiloint2double(t[3], t[2]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = (T)__int_as_float(t[e]);
    }
  }
}

template <typename T, int VEC>
__device__ __forceinline__ void team_loo_row(const RowsParams& P, const FastParams& F, TeamSmem& sm, const int64_t r,
                                             T (&v)[kTeamSlots], const T* rp_next) {
  constexpr int EPT = kTeamSlots;
  constexpr int NQ = EPT / VEC;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index in the team (uniform)
  const int S = P.n_draws;
  const int M = P.tail_count;
  const int gsz = F.gsz;
  const int kq = F.kq;
  const double* l1tab = sm.l1;
  const double* bgrid = sm.bg;
  const int mestM = F.mest_M;
  const double logS = F.log_S;
  const double INF = pinf();
  const int nvec = S / VEC;
  const int qfull = nvec / kTeamBlock;
  const int qrem = nvec - qfull * kTeamBlock;

  // ---- finish the load issued earlier: slots past the row copy this thread's first vector -------
#pragma unroll
  for (int q = 1; q < NQ; ++q) {
    if (q >= qfull) {
      const bool ok = (q == qfull) && (tid < qrem);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[q * VEC + e] = ok ? v[q * VEC + e] : v[e];
    }
  }
  // ---- 1. row statistics --------------------------------------------------------------------------
  double mx, mn, gs;
  {
    const T ninf = (T)(-INF);
    T cur = ninf, vmx = ninf, snap = ninf;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      cur = vmax_nc<true>(v[i], cur);
      vmx = vmax_nc<false>(v[i], vmx);
      if (i == gsz - 1) snap = cur;
    }
    mx = (double)cur;
    mn = -(double)vmx;
    gs = (double)snap;
  }
  const double wmx = wave_all<R_MAX>(mx);
  const double wmn = wave_all<R_MIN>(mn);
  // per-wave speculative threshold (see pla_wave.h): >= kq of this wave's 64 group maxima lie below it
  double wt;
  {
    double lo = wave_all<R_MIN>(gs), hi = wmx;
#pragma unroll 1
    for (int it = 0; it < 12; ++it) {
      const double mid = 0.5 * (lo + hi);
      const int below = __popcll(__ballot(gs < mid));
      if (below >= kq) hi = mid; else lo = mid;
    }
    wt = hi;
  }
  if (lane == 0) { sm.xw[w][0] = wmx; sm.xw[w][1] = wmn; sm.xw[w][2] = wt; }
  team_sync();  // (also: the previous row is done with every LDS array)
  const double m = fmax(sm.xw[0][0], sm.xw[1][0]);
  mn = fmin(sm.xw[0][1], sm.xw[1][1]);
  const double R = m - mn;
  const double t1 = 0.5 * (sm.xw[0][2] + sm.xw[1][2]) - m;  // mean of the two waves' estimates
  bool slow = !(R < kWaveMaxRange) || !(t1 < 0.0);
  const int k1 = key256(t1);
  const int kpad = key256(-R);
  if (kpad >= k1) slow = true;
  double khat = INF, loo = 0.0, lppd = 0.0;
  double s1 = 0.0, s2 = 0.0;
  unsigned ncand = 0;
  bool prefetched = false;
  if (!slow) {
    const int span = -k1;
    const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
    {
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(&sm.hist[4 * tid]) = z4;  // 128 threads x 4 bins = 512
    }
    const T padv = (T)(-mn);
#pragma unroll
    for (int q = 1; q < NQ; ++q) {
      if (q >= qfull) {
        const bool ok = (q == qfull && tid < qrem);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q * VEC + e] = ok ? v[q * VEC + e] : padv;
      }
    }
    // ---- 2. sweep over this thread's 32 draws; candidates go to this wave's list segment ----------
    const double* tab = sm.tab;
    double* seg = sm.cand + w * kTeamSeg;
    double* dump = sm.cand + 2 * kTeamSeg + tid;
    double magic = kMagic;
    asm volatile("" : "+v"(magic));
    constexpr int kPF = 3;
    double px[kPF], pt[kPF];
    double2 ptt[kPF];
#pragma unroll
    for (int i = 0; i < EPT + kPF; ++i) {
      if (i >= kPF) {
        const int sl = (i - kPF) % kPF;
        const double x = px[sl], t = pt[sl];
        const int k = __double2loint(t);
        const double rr = fma(t - magic, -kLn2_256, x);
        const double r2 = rr * rr;
        const double E = fma(fma(4.16666666666666666667e-02, r2, 0.5), r2, 1.0);
        const double O = fma(1.66666666666666666667e-01, r2, 1.0);
        const int es = (k << 12) & 0xfff00000;
        s1 = fma(add_hi(ptt[sl].x, es), fma(rr, O, E), s1);
        s2 = fma(add_hi(ptt[sl].y, -es), fma(-rr, O, E), s2);
        if ((i & 3) == 3) asm volatile("" : "+v"(s1), "+v"(s2));
      }
      if (i < EPT) {
        const int sl = i % kPF;
        const double x = (-(double)v[i]) - m;
        const double t = fma(x, kC256, magic);
        const int k = __double2loint(t);
        px[sl] = x;
        pt[sl] = t;
        ptt[sl] = *reinterpret_cast<const double2*>(tab + 2 * (k & 255));
        const bool cand = k >= k1;
        const unsigned long long cm = __ballot(cand);
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
        const unsigned pos = ncand + rank;
        double* dst = (cand && pos < (unsigned)kTeamSeg) ? (seg + pos) : dump;
        *dst = x;
        ncand += (unsigned)__popcll(cm);
      }
    }
    {  // pads
      const double x = -R;
      const double t = fma(x, kC256, magic);
      const int k = __double2loint(t);
      const double rr = fma(t - magic, -kLn2_256, x);
      const double2 tt = *reinterpret_cast<const double2*>(tab + 2 * (k & 255));
      const double r2 = rr * rr;
      const double E = fma(fma(4.16666666666666666667e-02, r2, 0.5), r2, 1.0);
      const double O = fma(1.66666666666666666667e-01, r2, 1.0);
      const int es = (k << 12) & 0xfff00000;
      const double npad = (double)((NQ - qfull) * VEC - ((tid < qrem) ? VEC : 0));
      s1 = fma(-npad * add_hi(tt.x, es), fma(rr, O, E), s1);
      s2 = fma(-npad * add_hi(tt.y, -es), fma(-rr, O, E), s2);
    }
    // the row registers are dead: stream the next row into them behind everything that follows
#ifndef PLA_TEAM_LATE_PREFETCH
    if (rp_next) issue_team_loads<T, VEC>(v, rp_next, S);
    prefetched = true;
#endif
    if (lane == 0) sm.xw[w][3] = (double)ncand;
    team_sync();
    const unsigned n0 = (unsigned)sm.xw[0][3], n1 = (unsigned)sm.xw[1][3];
    const unsigned nc = n0 + n1;
    if (n0 > (unsigned)kTeamSeg || n1 > (unsigned)kTeamSeg || (int)nc < M + 1) {
      slow = true;  // the speculative threshold missed
    } else {
      // list index -> LDS slot: segment 0 first, then segment 1
      auto at = [&](unsigned c) -> double { return sm.cand[c < n0 ? c : (kTeamSeg + c - n0)]; };
      // ---- 3. histogram of the candidates, suffix scan (both waves scan all 512 bins) -------------
      for (unsigned c = tid; c < nc; c += kTeamBlock) atomicAdd(&sm.hist[(key256(at(c)) - k1) >> sh], 1u);
      team_sync();
      int bstar = 0, C1 = 0;
      {
        unsigned c[8];
        unsigned tot = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint4 h = *reinterpret_cast<const uint4*>(&sm.hist[8 * lane + 4 * i]);
          c[4 * i] = h.x; c[4 * i + 1] = h.y; c[4 * i + 2] = h.z; c[4 * i + 3] = h.w;
          tot += h.x + h.y + h.z + h.w;
        }
        unsigned suf = tot;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const unsigned t = (unsigned)__shfl_down((int)suf, o);
          if (lane + o < kWave) suf += t;
        }
        unsigned a = suf - tot;
        int fb = -1, fc = 0;
        unsigned st[8];
#pragma unroll
        for (int i = 7; i >= 0; --i) {
          st[i] = a;
          if ((unsigned)M >= a && (unsigned)M < a + c[i]) {
            fb = 8 * lane + i;
            fc = (int)(a + c[i]);
          }
          a += c[i];
        }
        if (w == 0)
          *reinterpret_cast<uint4*>(&sm.start[8 * lane]) =
              make_uint4(st[0] | (st[1] << 16), st[2] | (st[3] << 16), st[4] | (st[5] << 16), st[6] | (st[7] << 16));
        const unsigned long long who = __ballot(fb >= 0);
        const int src = __ffsll((long long)who) - 1;
        bstar = __builtin_amdgcn_readlane(fb, src);
        C1 = __builtin_amdgcn_readlane(fc, src);
      }
      team_sync();
      if (C1 > kWaveCap) {
        slow = true;
      } else {
        // ---- 4. candidates at/above the boundary bin -> sa, grouped by bin --------------------------
        const int kstar = k1 + (bstar << sh);
        for (unsigned c = tid; c < nc; c += kTeamBlock) {
          const double x = at(c);
          const int k = key256(x);
          if (k >= kstar) {
            const int b = (k - k1) >> sh;
            const unsigned slot = sm.start[b] + (atomicSub(&sm.hist[b], 1u) - 1u);
            sm.sa[slot] = x;
          }
        }
        team_sync();
        double* sb = sm.cand;  // the lists are consumed: sorted candidates live here from now on
        // ---- 5. exact descending rank inside each bin -------------------------------------------------
        for (int c = tid; c < C1; c += kTeamBlock) {
          const double x = sm.sa[c];
          const int b = (key256(x) - k1) >> sh;
          const int lo = (int)sm.start[b];
          const int hi = (b > 0) ? (int)sm.start[b - 1] : C1;
          int cnt = 0;
          for (int c2 = lo; c2 < hi; ++c2) {
            const double x2 = sm.sa[c2];
            cnt += (x2 > x || (x2 == x && c2 > c)) ? 1 : 0;
          }
          sb[lo + cnt] = x;
        }
        team_sync();
        const double xcut = sb[M];
        int n = M;
        while (n > 0 && sb[n - 1] == xcut) --n;
        const double e_cut = exp_tab(xcut, sm.tab);
        double acc_t = 0.0, acc_r = 0.0;
        bool smoothed = false;
        if (n > 4) {
          team_sync();  // everybody has read sb[M..] before sa / sb are rewritten
          for (int j = tid; j < n; j += kTeamBlock) sm.sa[j] = exp_tab(sb[n - 1 - j], sm.tab) - e_cut;  // psis.py:147
          team_sync();
          const double* y = sm.sa;
          const double nn = (double)n;
          for (int p2 = tid; 2 * p2 + 1 < n; p2 += kTeamBlock) {
            const double2 yy = *reinterpret_cast<const double2*>(y + 2 * p2);
            *reinterpret_cast<double2*>(&sb[2 * p2]) = make_double2(yy.x + yy.y, yy.x * yy.y);
          }
          team_sync();
          const double* yp = sb;
          // ---- 6. GPD fit: lane j <-> grid point b_j; wave w multiplies its half of the factors -------
          const int mest = 30 + isqrt_i(n);
          const double yq = y[((n + 2) >> 2) - 1];
          const double yn = y[n - 1];
          const bool act = lane < mest;
          double b = (mest == mestM) ? bgrid[lane] : 1.0 - sqrt((double)mest / ((double)(lane + 1) - 0.5));
          b = div_fast(b, 3.0 * yq);
          b += recip_fast(yn);
          const double b_first = lane_value(b, 0);
          const double b_last = uniform_d(__shfl(b, mest - 1));
          const double fbig = fma(-b_first, yn, 1.0), fsmall = fma(-b_last, yn, 1.0);
          const bool wide = (fbig < 0x1p60) && (fsmall > 0x1p-60);
          const bool tiny = __ballot(act && fabs(b * yn) < 0.015625) != 0ull;
          // this wave's share of the factors: an even number of them, split at a multiple of 8
          const int half = ((n / 2 + 7) & ~7);
          const int i0 = (w == 0) ? 0 : (half < n ? half : n);
          const int i1 = (w == 0) ? (half < n ? half : n) : n;
          ProdAcc acc, acc2;
          acc.init();
          acc2.init();
          const double nb = -b;
          double corr = 0.0;
          int i = i0;
          if (wide && !tiny) {
            for (; i + 8 <= i1; i += 8) {
              const double2 pa = *reinterpret_cast<const double2*>(yp + i);
              const double2 pb = *reinterpret_cast<const double2*>(yp + i + 2);
              const double2 pc = *reinterpret_cast<const double2*>(yp + i + 4);
              const double2 pd = *reinterpret_cast<const double2*>(yp + i + 6);
              acc.mul(fma(nb, fma(nb, pa.y, pa.x), 1.0));
              acc2.mul(fma(nb, fma(nb, pb.y, pb.x), 1.0));
              acc.mul(fma(nb, fma(nb, pc.y, pc.x), 1.0));
              acc2.mul(fma(nb, fma(nb, pd.y, pd.x), 1.0));
              acc.renorm();
              acc2.renorm();
            }
            for (; i < i1; ++i) acc.mul(fma(nb, y[i], 1.0));
            acc.renorm();
          } else {
            for (; i < i1; ++i) {
              const double yi = y[i];
              const double f = fma(nb, yi, 1.0);
              corr += fma(nb, yi, 1.0 - f) * __builtin_amdgcn_rcp(f);
              acc.mul(f);
              acc.renorm();
            }
          }
          acc.m *= acc2.m;
          acc.e += acc2.e;
          double* part = sm.cand + 2 * kTeamSeg - 6 * kWave;  // last 384 doubles of the (dead) list segments
          part[(w * kWave + lane) * 3 + 0] = acc.m;
          part[(w * kWave + lane) * 3 + 1] = (double)acc.e;
          part[(w * kWave + lane) * 3 + 2] = corr;
          team_sync();
          const double pm = part[lane * 3] * part[(kWave + lane) * 3];
          const double pe = part[lane * 3 + 1] + part[(kWave + lane) * 3 + 1];
          const double pcorr = part[lane * 3 + 2] + part[(kWave + lane) * 3 + 2];
          const double rn = recip_fast(nn);
          const double kj = ((log_fast(pm) + pe * kLn2) + pcorr) * rn;                // psis.py:190
          const double ls = nn * (log_fast(-div_fast(b, kj)) - kj - 1.0);             // psis.py:191
          const double lmax = wave_all<R_MAX>(act ? ls : -INF);
          const bool anynan = (__ballot(act && (ls != ls)) != 0ull) || !(fabs(lmax) < INF);
          double wgt = act ? exp_neg(ls - lmax, sm.tab) : 0.0;                        // psis.py:192
          const double se = wave_all<R_SUM>(wgt);
          wgt = anynan ? qnan() : wgt * recip_fast(se);
          const bool keep = act && (wgt >= 10.0 * kEps);                              // psis.py:194-197
          const double sw = wave_all<R_SUM>(keep ? wgt : 0.0);
          const double bw = wave_all<R_SUM>(keep ? b * wgt : 0.0);
          const double b_post = (sw > 0.0) ? div_fast(bw, sw) : 0.0;                  // psis.py:198,201
          double pr = 1.0;
          for (int ii = tid; ii < n; ii += kTeamBlock) pr *= fma(-b_post, y[ii], 1.0);  // psis.py:203
          const double lsum = wave_all<R_SUM>(log_fast(pr));
          if (lane == 0) sm.xw[w][4] = lsum;
          team_sync();
          const double k_post = (sm.xw[0][4] + sm.xw[1][4]) * rn;
          const double sigma = -k_post / b_post;                                      // psis.py:205
          khat = (nn * k_post + 5.0) / (nn + 10.0);                                   // psis.py:206
          if (isfinite(khat)) {
            smoothed = true;
            const double rk = 1.0 / khat;
            const bool ktiny = fabs(khat) < kEps;
            for (int j = tid; j < n; j += kTeamBlock) {
              const double l1 = (n == M) ? l1tab[j] : log_fast(1.0 - ((double)j + 0.5) * rn);
              double q;
              if (sigma <= 0.0) {
                q = qnan();
              } else {
                q = ktiny ? -l1 : expm1_tab(-khat * l1, sm.tab) * rk;
                q *= sigma;
              }
              double wj = q + e_cut;
              if (wj > 1.0) wj = 1.0;
              const double ej = y[j] + e_cut;
              acc_t += wj - ej;
              acc_r += div_fast(wj, ej);
            }
          }
        }
        // ---- 7. combine the two waves' sums ------------------------------------------------------------
        const double tw = wave_all<R_SUM>(s1 + acc_t);
        const double s2w = wave_all<R_SUM>(s2);
        const double arw = wave_all<R_SUM>(acc_r);
        if (lane == 0) { sm.xw[w][5] = tw; sm.xw[w][6] = s2w; sm.xw[w][7] = arw; }
        team_sync();
        const double total = sm.xw[0][5] + sm.xw[1][5];
        const double s2t = sm.xw[0][6] + sm.xw[1][6];
        double tail_ratio = (double)S;
        if (smoothed) tail_ratio = (double)(S - n) + (sm.xw[0][7] + sm.xw[1][7]);
        loo = log_fast(div_fast(tail_ratio, total)) - m;
        lppd = (log_fast(s2t) - R) + ((-mn) - logS);
        if (!(total > 1e-280) || !isfinite(loo) || !isfinite(lppd)) slow = true;
      }
    }
  }
  if (!prefetched && rp_next) issue_team_loads<T, VEC>(v, rp_next, S);
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
}

#ifndef PLA_TEAM_WPS
#define PLA_TEAM_WPS 4
#endif
template <typename T, int VEC>
__global__ __launch_bounds__(kTeamBlock, PLA_TEAM_WPS) void team_loo_kernel(RowsParams P, FastParams F) {
  __shared__ __attribute__((aligned(16))) TeamSmem sm;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < kTabN / kTeamBlock; ++i) {
    const int j = tid + kTeamBlock * i;
    sm.tab[2 * j] = exp2((double)j * (1.0 / kTabN));
    sm.tab[2 * j + 1] = exp2(-(double)j * (1.0 / kTabN));
  }
  for (int j = tid; j < P.tail_count; j += kTeamBlock) sm.l1[j] = F.l1_table[j];
  if (tid < kWave) sm.bg[tid] = F.b_grid[tid];
  __syncthreads();
  T v[kTeamSlots];
  const T* base = reinterpret_cast<const T*>(P.in);
  if ((int64_t)blockIdx.x < P.n_obs) issue_team_loads<T, VEC>(v, base + (int64_t)blockIdx.x * P.stride_obs, P.n_draws);
  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const int64_t rn = r + gridDim.x;
    team_loo_row<T, VEC>(P, F, sm, r, v, rn < P.n_obs ? base + rn * P.stride_obs : nullptr);
  }
}

}  // namespace pla
