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

#ifndef PLA_COL_U
#define PLA_COL_U 12       // draws per batch; two batches in flight per lane
#endif
#ifndef PLA_COL_WAVES
#define PLA_COL_WAVES 2    // waves per SIMD the sweep is compiled for (its staging rings allow two workgroups per CU)
#endif
#ifndef PLA_COL_ILP
#define PLA_COL_ILP 4      // draws the scheduler may interleave
#endif
#ifndef PLA_COL_STORE_AUX
#define PLA_COL_STORE_AUX 0   // cache policy of the candidate stores
#endif
#ifndef PLA_COL_FLUSH
#define PLA_COL_FLUSH 16      // candidates per store burst: 16 x 8 bytes = ONE cache line, written whole
#endif
#ifndef PLA_COL_CHECK
#define PLA_COL_CHECK PLA_COL_U  // draws between two looks at the staging ring: once per batch (a flush is eight store instructions whether or not a lane has a burst ready)
#endif
constexpr int kColFlush = PLA_COL_FLUSH;
static_assert(PLA_COL_FLUSH - 1 + PLA_COL_CHECK <= 2 * PLA_COL_FLUSH, "the ring must hold what arrives between two checks");
constexpr int kColRing = 2 * kColFlush;  // staging ring per lane (at most kColFlush - 1 + PLA_COL_CHECK entries wait between two checks)
constexpr int kColSample = 512;   // draws in the pre-pass: 64 groups of 8
constexpr int kColCap = 1024;     // candidate list capacity per observation (doubles)
// Layout of the lists: 64 consecutive observations (the lanes of one sweep wave) share a 512 KB group, interleaved line by
// line -- entry j of observation 64 g + l lies at double  g * 65536 + ((j / 16) * 64 + l) * 16 + j % 16.  A flush writes one
// WHOLE 128-byte line (16 candidates), so no line is ever written in two halves: half-line bursts into these lines took the
// sweep from 2.0 to 4.7 ms per block (read-modify-write), half-line bursts interleaved per 64 bytes -- the two halves of a
// line belong to neighbouring lanes -- 2.04 ms, whole lines 1.90 ms although the 32-entry rings leave room for two
// workgroups per CU, not three.  The lines the lanes of a wave write at about the same time are neighbours (a few pages per
// store instruction instead of 64 pages 8 KB apart), and the selection reads whole lines.
__device__ __forceinline__ int64_t col_list_base(int64_t obs) { return (obs >> 6) * (int64_t)(64 * kColCap) + (obs & 63) * 16; }
__device__ __forceinline__ int col_list_entry(int j) { return (j >> 4) * 1024 + (j & 15); }

struct ColParams {
  const void* in;      // element (observation i, draw s) at in[s * ld + i]
  int64_t n_obs;       // observations of this launch
  int n_draws;
  int64_t ld;          // elements between consecutive draws
  int kq;              // the threshold has kq of the 64 group maxima below it
  double* cand;        // [n_obs][kColCap] log-likelihoods of the draws whose raw = -ll is at or above the threshold
  double* scal;        // [n_obs][8]: m', max raw, min raw, sum e^x', sum e^-x', number of candidates (uncapped), threshold, -
};

// draw `u` of a batch: the batch's first draw is the base of the descriptor, u * ld * sizeof(T) rides in a scalar offset and
// the lane's observation in the ONE vector offset all loads of the kernel share -- no 64-bit address per load in vector
// registers (sixteen of them cost 32 registers and spilled)
template <typename T>
__device__ __forceinline__ T col_load(const __amdgpu_buffer_rsrc_t rs, const int voff, const int soff) {
  if constexpr (sizeof(T) == 8) {
    typedef int v2i __attribute__((ext_vector_type(2)));
    const v2i t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, PLA_LOAD_AUX);
    return (T)__hiloint2double(t[1], t[0]);
  } else {
    return (T)__int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, PLA_LOAD_AUX));
  }
}

template <typename T>
__global__ __launch_bounds__(256, PLA_COL_WAVES) void col_sweep_kernel(ColParams P) {
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  // candidates are staged here, per lane, and leave for the workspace kColFlush at a time.  (Appending them one by one --
  // 8-byte stores, each to another cache line, ten draws apart -- kept 262 144 partly written lines open against 4 MB of
  // L2 per XCD and doubled the kernel's run time.)  Per lane: a ring of 2 kColFlush entries + a dump slot for draws that are
  // no candidates (so that the append needs no branch) + one entry of padding that keeps the 16-byte alignment.
  __shared__ __attribute__((aligned(16))) double stage[256][kColRing + 2];
  for (int j = threadIdx.x; j < kTabN; j += 256) exp_table_entry(tab, j);
  __syncthreads();
  const int S = P.n_draws;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < P.n_obs;
  const T* col = reinterpret_cast<const T*>(P.in) + (live ? i : P.n_obs - 1);  // (idle lanes re-read the last observation)
  const int voff = (int)((live ? i : P.n_obs - 1) * (int64_t)sizeof(T));            // this lane's byte offset inside a draw
  const int64_t draw_bytes = P.ld * (int64_t)sizeof(T);
  const double INF = pinf();

  // ---- pre-pass: 64 group maxima over 512 draws spread over the row; group g holds the samples g, g + 64, ... ----------
  float gm[64];
  double mp = -INF;  // provisional shift: the largest sampled raw value
#pragma unroll
  for (int g = 0; g < 64; ++g) gm[g] = -__builtin_inff();
#pragma unroll 1
  for (int k = 0; k < kColSample / 64; ++k) {
#pragma unroll
    for (int g0 = 0; g0 < 64; g0 += 16) {  // 16 loads in flight at a time (64 would need 128 registers for the draws alone)
      T v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = col[(((int64_t)(k * 64 + g0 + j) * S) / kColSample) * P.ld];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double raw = -(double)v[j];
        mp = fmax(mp, raw);
        // rounded up: the threshold may only err towards FEWER candidates by what one float ulp is worth
        gm[g0 + j] = fmaxf(gm[g0 + j], __double2float_ru(raw));
      }
      __builtin_amdgcn_sched_barrier(0);
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

  // ---- the pass: every draw of the observation ------------------------------------------------------------------------
  double nmn = INF, nmx = -INF, s1 = 0.0, s2 = 0.0;  // min / max of ll = -(max / min of raw)
  int cnt = 0, flushed = 0;  // candidates seen / written to the workspace (a multiple of kColFlush)
  double* const mine = stage[threadIdx.x];
  // (the lists of this launch's observations: at most 262 144 x 8 KB = 2 GB, inside one descriptor's 32-bit range)
  const __amdgpu_buffer_rsrc_t rs_list = __builtin_amdgcn_make_buffer_rsrc(
      P.cand, 0, (int)(unsigned)(((((P.n_obs + 63) & ~63ll) * (int64_t)kColCap * 8) > 0xfffffff0ll) ? 0xfffffff0ll : ((P.n_obs + 63) & ~63ll) * (int64_t)kColCap * 8), 0x00020000);
  static_assert(kColFlush == 16, "a flush is one whole line of the interleaved list layout");
  const int list_off = (int)(col_list_base(live ? i : 0) * 8);
  const char* tabc = reinterpret_cast<const char*>(tab);
  constexpr int U = PLA_COL_U;
  int c4096 = 4096, cm4096 = -4096, four = 4;
  asm volatile("" : "+s"(c4096), "+s"(cm4096));
  asm volatile("" : "+v"(four));
  // (`ll` = the stored log-likelihood; raw = -ll is never materialised: the negation rides in the instructions' operand
  // modifiers, and the candidate lists hold ll as well)
  const double nmp = -mp, nt_raw = -t_raw;
  const auto one = [&](double ll) {
    nmn = fmin(nmn, ll);                             // -max raw
    nmx = fmax(nmx, ll);                             // -min raw
    const double x = nmp - ll;                       // raw - m': psis.py:134 about the provisional shift
    const double t = fma(x, kC256, kMagic);
    const int k = __double2loint(t);                 // round(x * 256 / ln 2)
    const int4 tt = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));  // 16 * (k & 255)
    const double rr = fma(t - kMagic, -kLn2_256, x);
    const double r2 = rr * rr;
    const double E = fma(r2, 0.5, 1.0);              // cosh rr to 1.5e-13 (as in the wave kernel's sweep)
    const double O = fma(1.66666666666666666667e-01, r2, 1.0);
    s1 = fma(__hiloint2double(mad_i24(k, c4096, tt.y), tt.x), fma(rr, O, E), s1);
    s2 = fma(__hiloint2double(mad_i24(k, cm4096, tt.w), tt.z), fma(-rr, O, E), s2);
    // candidate: into this lane's staging ring, without control flow (everything else goes to the lane's dump slot)
    const bool cand = ll <= nt_raw;                  // raw >= threshold
    mine[cand ? (cnt & (kColRing - 1)) : kColRing] = ll;
    cnt += cand ? 1 : 0;
  };
  // kColFlush staged candidates -> the observation's list, when there are that many (called every PLA_COL_CHECK-th draw: at most that many
  // arrive in between, so the ring never overflows).  Bounds-checked buffer stores: what must not be written -- nothing to
  // flush yet, list full, idle lane -- gets an offset past the end of the descriptor, which the hardware drops.
  typedef int v4i __attribute__((ext_vector_type(4)));
  const auto flush4 = [&]() {
    const bool go = cnt - flushed >= kColFlush;
    const bool wr = go & (flushed + kColFlush <= kColCap) & live;  // (bitwise: no short-circuit branches inside the sweep)
    const int off = wr ? list_off + 8 * col_list_entry(flushed) : (int)0xffffff00;  // (flushed is a multiple of 16: the start of a line)
    const double* src = &mine[flushed & (kColRing - 1)];  // (flushed is a multiple of kColFlush: no wrap inside a burst)
#pragma unroll
    for (int q = 0; q < kColFlush / 2; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4i*>(src + 2 * q), rs_list, off, 16 * q, PLA_COL_STORE_AUX);
    flushed += go ? kColFlush : 0;
  };
  // Two batches of U draws per lane: the loads of batch b + 1 are in flight while batch b is computed (each wave keeps
  // 2 U x 512 bytes of HBM reads outstanding: with ~12 waves per CU that is the 100+ KB per CU the HBM latency asks for --
  // with one batch of 8 the kernel ran at 2.8 TB/s, waiting 5 us for every 0.5 us of arithmetic)
  T buf[2][U];
  const int nb = S / U;  // whole batches
  const auto fetch = [&](T (&dst)[U], int b) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(P.in) + (int64_t)b * U * draw_bytes), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u] = col_load<T>(rs, voff, (int)(u * draw_bytes));
  };
  const auto work = [&](const T (&src)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      one((double)src[u]);
      if (u % PLA_COL_CHECK == PLA_COL_CHECK - 1) flush4();
      // (a few draws interleave -- the table read of one under the arithmetic of the others -- but not all sixteen: the
      // scheduler would otherwise start every draw of the batch at once and spill)
      if (u % PLA_COL_ILP == PLA_COL_ILP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (nb > 0) fetch(buf[0], 0);
  int b = 0;
#pragma unroll 1
  for (; b + 2 <= nb; b += 2) {
    fetch(buf[1], b + 1);
    work(buf[0]);
    if (b + 2 < nb) fetch(buf[0], b + 2);
    work(buf[1]);
  }
  if (b < nb) work(buf[0]);  // an odd last batch (already fetched)
  for (int s = nb * U; s < S; ++s) {
    one((double)col[(int64_t)s * P.ld]);
    flush4();
  }
  flush4();
  {  // the last, partly filled line goes out whole as well (entries past `cnt` are stale ring contents, never read): eight
     // byte-sized stores into a line that is not complete would each be a read-modify-write
    const bool wr = (cnt > flushed) & (flushed + kColFlush <= kColCap) & live;
    const int off = wr ? list_off + 8 * col_list_entry(flushed) : (int)0xffffff00;
    const double* src = &mine[flushed & (kColRing - 1)];
#pragma unroll
    for (int q = 0; q < kColFlush / 2; ++q)
      __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4i*>(src + 2 * q), rs_list, off, 16 * q, PLA_COL_STORE_AUX);
  }
  if (live) {
    double* o = P.scal + i * 8;
    o[0] = mp; o[1] = -nmn; o[2] = -nmx; o[3] = s1; o[4] = s2; o[5] = (double)cnt; o[6] = t_raw;
  }
}

// One wavefront per observation: the split pass's selection on the candidate list, read where it lies (no LDS copy: the
// scratch per wave is the histogram, the bin offsets and the bin-grouped tail: 6 KB, so 24 waves fit a CU and hide the
// latencies this phase consists of).  Workgroups of 4 independent waves sharing the exponential table.
template <class CAP>
struct ColSmem {
  using Caps = CAP;
  unsigned hist[kWaveBins];
  unsigned short start[kWaveBins];
  double sa[CAP::kSa + 4];
  double dump_slot[kWave];
  unsigned dump_bin[kWave];
};
template <class SM>
struct CandInList {  // x = raw - max raw with the reference's single rounding (psis.py:134); the list holds ll = -raw
  const double* list;
  double m;
  SM& sm;
  __device__ __forceinline__ double at(unsigned c) const { return (-list[col_list_entry((int)c)]) - m; }
  __device__ __forceinline__ unsigned* dump_bin(int lane) const { return &sm.dump_bin[lane]; }
  __device__ __forceinline__ double* dump_slot(int lane) const { return &sm.dump_slot[lane]; }
};

template <class CAP>
__global__ __launch_bounds__(kWave * 4) void col_select_kernel(ColParams P, FastParams F, int tail_count) {
  using SM = ColSmem<CAP>;
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
    const double s1p = uniform_d(sc[3]), s2p = uniform_d(sc[4]), t_raw = uniform_d(sc[6]);
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
      double magic = kMagic, c256 = kC256;
      // histogram origin: every candidate has raw >= t_raw, so x >= t_raw - m up to the two roundings: two keys of margin
      const int k1 = __double2loint(fma(t_raw - m, c256, magic)) - 2;
      const int span = -k1;
      const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
      // the sums about the true shift: e^x = e^x' e^-(m - m'), e^-x = e^-x' e^(m - m')   (R < 690 keeps both finite)
      const double s1 = lane == 0 ? s1p * exp_tab(-delta, tb.tab) : 0.0;
      const double s2 = lane == 0 ? s2p * exp_tab(delta, tb.tab) : 0.0;
      const CandInList<SM> src{P.cand + col_list_base(r), m, sm};
      wave_sync();
      wave_select_split<SM, TB, (CAP::kMaxTail + 63) / 64, CandInList<SM>>(F, sm, tb, r, lane, M, m, mn, s1, s2, (unsigned)ncand, k1, sh,
                                                                           magic, c256, slow, src);
    }
    if (slow && lane == 0) {
      const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
      F.slow_list[idx] = (unsigned)r;
      F.ws_s[r * F.ws_sstride + 5] = -1.0;  // tail length -1: on the list, nothing for the fit kernel
    }
  }
}

}  // namespace pla
