// PSIS-LOO for observations-fastest log-likelihood matrices (the layout `pl.loo(idata)` hands over, loo.py:189), second form:
// a WORKGROUP PER 16 NEIGHBOURING OBSERVATIONS, candidate lists in LDS.
//
// pla_col.h gives every observation a lane and must therefore keep the candidate lists in HBM (8 KB per observation written
// and read again: 1.5 of its 7.3 ms sweep, plus a separate selection kernel that is all latency) and re-reads a 12.8 % sample of
// the matrix from HBM for its threshold.  Here the unit is the 128-byte piece of a draw that 16 neighbouring observations share
// (tools/microbench/col_stream.hip: pieces of that size stream at the same 6.5 TB/s as the 512-byte pieces of the lane-per-
// observation sweep): one workgroup of eight waves owns the 16 observations, lane = (observation l & 15, draw l >> 4), a wave
// load reads the pieces of four consecutive draws, the eight waves cover 32 draws per step.  32 lanes work on every
// observation, so its candidates fit the CU: 16 lists of 640 entries in LDS, appended to with LDS atomics, and the selection
// of the split pass (wave_select_split, pla_wave.h) runs on them in place, two observations per wave -- nothing but the
// hand-over to fit_rows_kernel goes back to HBM.  The threshold sample (512 draws spread over the row, 16 per lane) is read a
// second time by the sweep, but the 16 MB of samples the 256 resident workgroups hold at a time never leave the Infinity Cache.
//
//   per group of 16 observations:
//   A  sample: 16 draws per lane, spread over the row (every chain of a chain-major stack contributes), as f32 keys to LDS
//   B  threshold: the sample's ks-th largest value, by bisection on exact counts (one wave per observation, ballots)
//   C  sweep: every draw once -- max / min, the two sums of e^x', e^-x' about the provisional shift (the sample's maximum), and
//      the draws at or above the threshold appended to the observation's list
//   D  selection on the list with the true shift (psis.py:134: x = raw - max raw in one rounding), hand-over to the fit kernel
#pragma once

#include "pla_col.h"

namespace pla {

#ifndef PLA_TILE_U
#define PLA_TILE_U 12      // steps per batch; two batches in flight per lane
#endif
#ifndef PLA_TILE_CAP
#define PLA_TILE_CAP 640   // candidate list capacity per observation
#endif
#ifndef PLA_TILE_ILP
#define PLA_TILE_ILP 4     // draws the scheduler may interleave
#endif
constexpr int kTileWaves = 8;
constexpr int kTileThreads = kWave * kTileWaves;
constexpr int kTileSample = 512;
constexpr int kTileCap = PLA_TILE_CAP;
constexpr int kTileBisect = 16;

struct TileParams {
  const void* in;      // element (observation i, draw s) at in[s * ld + i]
  int64_t n_obs;       // observations of this launch
  int n_draws;
  int64_t ld;          // elements between consecutive draws
  int ks;              // the threshold leaves at most ks of the 512 sampled draws at or above it
};

template <typename T>
struct TileSmem {
  static constexpr int kObs = 128 / (int)sizeof(T);  // observations per group: one 128-byte piece of a draw
  using Sel = ColSmem<CapsSmall>;
  double tab[2 * kTabN];
  union {
    T list[kObs][kTileCap];               // C, D: the candidates' stored log-likelihoods
    float keys[kObs][kTileSample + 1];    // A, B: the sample, raw = -ll rounded up to f32 (+1: the rows start in different banks)
  };
  Sel sel[kTileWaves];                    // D: scratch of one selection per wave
  double red[kTileWaves][kObs][4];        // C -> D: per wave and observation: min ll, max ll, sum e^x', sum e^-x'
  double scal[kObs][2];                   // B -> C, D: provisional shift, threshold (both raw)
  unsigned cnt[kObs];                     // candidates seen (not capped)
  unsigned dumpc[kTileThreads];           // where the lanes whose draw is no candidate count (from 2^31 up: "past the end of any list")
  T dumpv[kTileThreads];                  // ... and store
};

template <typename T>
struct CandInTile {  // x = raw - max raw with the reference's single rounding (psis.py:134); the list holds ll = -raw
  const T* list;
  double m;
  ColSmem<CapsSmall>& sm;
  __device__ __forceinline__ double at(unsigned c) const { return (-(double)list[c]) - m; }
  __device__ __forceinline__ unsigned* dump_bin(int lane) const { return &sm.dump_bin[lane]; }
  __device__ __forceinline__ double* dump_slot(int lane) const { return &sm.dump_slot[lane]; }
};

template <typename T>
__global__ __launch_bounds__(kTileThreads, 1) void tile_loo_kernel(TileParams P, FastParams F, int tail_count) {
  static_assert(sizeof(T) == 8, "lane mapping and list size are laid out for 8-byte draws");
  using SMT = TileSmem<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  SMT& sm = *reinterpret_cast<SMT*>(tile_lds);
  constexpr int kObs = SMT::kObs;                   // 16
  constexpr int kSub = kWave / kObs;                // draws per wave load: 4
  constexpr int kStep = kSub * kTileWaves;          // draws per step of the workgroup: 32
  constexpr int kPer = kTileSample / kStep;         // sampled draws per lane: 16
  constexpr int kSelPer = kObs / kTileWaves;        // observations a wave selects for: 2
  constexpr int U = PLA_TILE_U;
  const int tid = (int)threadIdx.x;
  for (int j = tid; j < kTabN; j += kTileThreads) exp_table_entry(sm.tab, j);
  sm.dumpc[tid] = 0x80000000u;
  __syncthreads();
  const int w = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int lane = wave_lane();
  const int o = lane & (kObs - 1), dsub = lane / kObs;
  const int S = P.n_draws, M = tail_count;
  const int64_t db = P.ld * (int64_t)sizeof(T);      // bytes between consecutive draws
  const int64_t step_bytes = db * kStep;
  const int nit = S / kStep;                         // whole steps
  const int nb = nit / U, left = nit - nb * U;       // whole batches, steps behind them
  const int64_t ngroups = (P.n_obs + kObs - 1) / kObs;
  const double INF = pinf();
  const char* tabc = reinterpret_cast<const char*>(sm.tab);

  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t obs0 = g * kObs;
    // (lanes past the last observation of the launch re-read the last one; nothing of theirs is selected)
    const int oc = (obs0 + o < P.n_obs) ? o : (int)(P.n_obs - 1 - obs0);
    // this lane inside a step: its observation and its draw of the wave's kSub (32-bit: ld < 2^32 / (8 kSub), checked by the launcher)
    const int voff = (int)((unsigned)(oc * (int)sizeof(T)) + (unsigned)dsub * (unsigned)db);
    // the wave's first piece of step 0 (scalar)
    const char* gbase = reinterpret_cast<const char*>(P.in) + obs0 * (int64_t)sizeof(T) + (int64_t)(kSub * w) * db;
    const auto load_step = [&](int step) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gbase + (int64_t)step * step_bytes), 0, (int)0xfffffff0u, 0x00020000);
      return col_load<T>(rs, voff, 0);
    };

    // ---- A. the sample: step floor((j + w / 8) nit / 16) for j < 16 -- 128 clusters of four consecutive draws, evenly spread -----
    {
      T sv[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) sv[j] = load_step((int)(((int64_t)(kTileWaves * j + w) * nit) / (kTileWaves * kPer)));
#pragma unroll
      for (int j = 0; j < kPer; ++j)  // rounded up: the threshold may only err towards FEWER candidates by what one float ulp is worth
        sm.keys[o][(w * kSub + dsub) * kPer + j] = __double2float_ru(-(double)sv[j]);
    }
    __syncthreads();
    // ---- B. threshold and provisional shift, one wave per observation -------------------------------------------------------
#pragma unroll 1
    for (int q = 0; q < kSelPer; ++q) {
      const int oo = w * kSelPer + q;
      float kx[kTileSample / kWave];
      float lmin = __builtin_inff(), lmax = -__builtin_inff();
#pragma unroll
      for (int k = 0; k < kTileSample / kWave; ++k) {
        kx[k] = sm.keys[oo][lane + kWave * k];
        lmin = fminf(lmin, kx[k]);
        lmax = fmaxf(lmax, kx[k]);
      }
      double hmax, nlmin;
      wave_all2<R_MAX>((double)lmax, -(double)lmin, hmax, nlmin);
      float lo = (float)(-nlmin), hi = (float)hmax;
      const int nbt = kTileSample - P.ks;  // a value with at least this many of the sample below it
#pragma unroll 1
      for (int it = 0; it < kTileBisect; ++it) {
        const float mid = 0.5f * (lo + hi);
        int below = 0;
#pragma unroll
        for (int k = 0; k < kTileSample / kWave; ++k) below += __popcll(__ballot(kx[k] < mid));
        if (below >= nbt) hi = mid;
        else lo = mid;
      }
      if (lane == 0) {
        sm.scal[oo][0] = hmax;
        sm.scal[oo][1] = (double)hi;
        sm.cnt[oo] = 0u;
      }
    }
    __syncthreads();

    // ---- C. the sweep ---------------------------------------------------------------------------------------------------------
    const double nmp = -sm.scal[o][0], nt_raw = -sm.scal[o][1];
    double nmn = INF, nmx = -INF, s1 = 0.0, s2 = 0.0;  // min / max of ll = -(max / min of raw)
    T* const mylist = sm.list[o];
    unsigned* const mycnt = &sm.cnt[o];
    unsigned* const mydumpc = &sm.dumpc[tid];
    T* const mydumpv = &sm.dumpv[tid];
    int c4096 = 4096, cm4096 = -4096, four = 4;
    asm volatile("" : "+s"(c4096), "+s"(cm4096));
    asm volatile("" : "+v"(four));
    // Software pipeline, kPF draws deep and carried from batch to batch: stage A of a draw (shift, range reduction, table
    // read, the request for a list slot) is issued kPF draws before its stage B (polynomial, accumulation, the store to the
    // slot), so that the LDS round trips of the table read and of the counter are covered by the arithmetic in between.
    constexpr int kPF = 3;
    static_assert(U % kPF == 0, "pipeline slots are assigned at compile time");
    double px[kPF], pt[kPF], pll[kPF];
    int4 ptt[kPF];
    unsigned ppos[kPF];
    const auto stage_a = [&](const double ll, const int sl) {
      nmn = fmin(nmn, ll);
      nmx = fmax(nmx, ll);
      const double x = nmp - ll;                       // raw - m': psis.py:134 about the provisional shift
      const double t = fma(x, kC256, kMagic);
      const int k = __double2loint(t);
      ptt[sl] = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));
      // candidate: a slot of the observation's list from its counter (32 lanes in 8 waves append to one list); everybody else
      // counts on a private counter that starts past the end of any list, and stores to a private slot: no control flow
      const bool cand = ll <= nt_raw;                  // raw >= threshold
      ppos[sl] = __hip_atomic_fetch_add(cand ? mycnt : mydumpc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      px[sl] = x;
      pt[sl] = t;
      pll[sl] = ll;
    };
    const auto stage_b = [&](const int sl) {
      const double x = px[sl], t = pt[sl];
      const int k = __double2loint(t);
      const double rr = fma(t - kMagic, -kLn2_256, x);
      const double r2 = rr * rr;
      const double E = fma(r2, 0.5, 1.0);              // cosh rr to 1.5e-13 (as in the wave kernel's sweep)
      const double O = fma(1.66666666666666666667e-01, r2, 1.0);
      s1 = fma(__hiloint2double(mad_i24(k, c4096, ptt[sl].y), ptt[sl].x), fma(rr, O, E), s1);
      s2 = fma(__hiloint2double(mad_i24(k, cm4096, ptt[sl].w), ptt[sl].z), fma(-rr, O, E), s2);
      *(ppos[sl] < (unsigned)kTileCap ? &mylist[ppos[sl]] : mydumpv) = (T)pll[sl];
    };
    const auto one = [&](double ll) {  // (outside the batches: the few draws behind them)
      stage_a(ll, 0);
      stage_b(0);
    };
    {
      T buf[2][U];
      const auto fetch = [&](T (&dst)[U], int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) dst[u] = load_step(b * U + u);
      };
      bool primed = false;  // (known at compile time at every call: the first batch has no stage B for its first kPF draws)
      const auto work = [&](const T (&src)[U], const bool first) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!(first && u < kPF)) stage_b(u % kPF);
          stage_a((double)src[u], u % kPF);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      const auto drain = [&]() {
#pragma unroll
        for (int u = 0; u < kPF; ++u) stage_b(u);
      };
      if (nb > 0) {
        fetch(buf[0], 0);
        if (nb > 1) fetch(buf[1], 1);
        work(buf[0], true);  // (the only copy of the loop body without the first stage Bs)
        primed = true;
      }
      int b = 1;
#pragma unroll 1
      for (; b + 2 <= nb; b += 2) {
        fetch(buf[0], b + 1);
        work(buf[1], false);
        if (b + 2 < nb) fetch(buf[1], b + 2);
        work(buf[0], false);
      }
      if (b < nb) work(buf[1], false);  // an even number of batches: the last one is already fetched
      if (primed) drain();
      // the steps behind the last whole batch, and the draws behind the last whole step
      if (left > 0) {
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < left) buf[0][u] = load_step(nb * U + u);
#pragma unroll
        for (int u = 0; u < U - 1; ++u)
          if (u < left) one((double)buf[0][u]);
      }
      if (nit * kStep + kSub * w + dsub < S) one((double)load_step(nit));
    }
    // ---- the 32 lanes of an observation: the four of a wave here, the eight waves in LDS ---------------------------------------
#pragma unroll
    for (int off = kObs; off < kWave; off *= 2) {
      nmn = fmin(nmn, __shfl_xor(nmn, off));
      nmx = fmax(nmx, __shfl_xor(nmx, off));
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    if (dsub == 0) {
      double* rd = sm.red[w][o];
      rd[0] = nmn; rd[1] = nmx; rd[2] = s1; rd[3] = s2;
    }
    __syncthreads();

    // ---- D. selection: wave w takes the observations kSelPer w + q ---------------------------------------------------------------
#pragma unroll 1
    for (int q = 0; q < kSelPer; ++q) {
      const int oo = w * kSelPer + q;
      const int64_t r = obs0 + oo;
      if (r >= P.n_obs) break;  // (wave-uniform)
      double lmin = INF, lmax = -INF, s1p = 0.0, s2p = 0.0;
#pragma unroll
      for (int k = 0; k < kTileWaves; ++k) {  // (every lane reads the same eight entries: broadcasts, and one order of summation)
        const double* rd = sm.red[k][oo];
        lmin = fmin(lmin, rd[0]);
        lmax = fmax(lmax, rd[1]);
        s1p += rd[2];
        s2p += rd[3];
      }
      const double m = uniform_d(-lmin), mn = uniform_d(-lmax);
      s1p = uniform_d(s1p);
      s2p = uniform_d(s2p);
      const double mp = uniform_d(sm.scal[oo][0]), t_raw = uniform_d(sm.scal[oo][1]);
      const int ncand = __builtin_amdgcn_readfirstlane((int)sm.cnt[oo]);
      const double R = m - mn, delta = m - mp;  // (delta: the row's maximum against the sample's, rounded up to f32)
      bool slow = !(R < kWaveMaxRange) || ncand < M + 1 || ncand > kTileCap || !(fabs(s1p) < INF) || !(fabs(s2p) < INF) ||
                  !(fabs(delta) < kWaveMaxRange);
      typename SMT::Sel& ss = sm.sel[w];
      if (!slow) {
        wave_sync();  // the previous observation is done with the scratch
        {
          const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int j = 0; j < kWaveBins / (4 * kWave); ++j) *reinterpret_cast<uint4*>(&ss.hist[4 * (lane + kWave * j)]) = z4;
        }
        double magic = kMagic, c256 = kC256;
        // histogram origin: every candidate has raw >= t_raw, so x >= t_raw - m up to the two roundings: two keys of margin
        const int k1 = __double2loint(fma(t_raw - m, c256, magic)) - 2;
        const int span = -k1;
        const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
        // the sums about the true shift: e^x = e^x' e^-(m - m'), e^-x = e^-x' e^(m - m')
        const double s1 = lane == 0 ? s1p * exp_tab(-delta, sm.tab) : 0.0;
        const double s2 = lane == 0 ? s2p * exp_tab(delta, sm.tab) : 0.0;
        const CandInTile<T> src{sm.list[oo], m, ss};
        wave_sync();
        wave_select_split<typename SMT::Sel, SMT, (CapsSmall::kMaxTail + 63) / 64, CandInTile<T>>(
            F, ss, sm, r, lane, M, m, mn, s1, s2, (unsigned)ncand, k1, sh, magic, c256, slow, src);
      }
      if (slow && lane == 0) {
        const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
        F.slow_list[idx] = (unsigned)r + F.slow_base;
        F.ws_s[r * F.ws_sstride + 5] = -1.0;  // tail length -1: on the list, nothing for the fit kernel
      }
    }
    __syncthreads();  // the lists are free for the next group's sample
  }
}

}  // namespace pla
