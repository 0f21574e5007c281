// Sixteen-lanes-per-observation GPD fit, smoothing sums and outputs: the second half of the split LOO pass.
//
// The wave kernel (pla_wave.h) is wave-per-observation because a row of S draws needs the whole register
// file; but everything after the exact selection works on <= 256 tail values per observation, and there a
// wavefront per observation wastes most of its lanes (21 of 64 in the grid, 63 of 64 in every scalar step,
// six-step cross-lane reductions everywhere).  In the split pass the wave kernel stops after the selection
// and hands over, per observation, the ascending tail values y_j (psis.py:146-147) and six scalars; this
// kernel gives every DPP row of 16 lanes one observation (four observations per wavefront):
//   * the tail sits in the registers of its 16 lanes (4 NQ values each, one coalesced read);
//   * (PLA_FIT_SORTS, the default) the wave kernel hands the tail over only GROUPED BY BIN of its selection histogram, bins
//     descending; the order inside the bins (a handful of values each) is finished here: the values are brought to a blocked
//     layout through LDS (lane t holds the descending ranks 4 NQ t .. 4 NQ t + 4 NQ - 1) and a few odd-even transposition
//     passes -- register min / max pairs, one DPP exchange per lane boundary -- run until every row of the wave is sorted
//     (checked, so any order, even a fully unsorted one, ends sorted).  That is ~35 issue slots per observation against
//     the ~420 the exact in-bin ranking costs the wave kernel, where 64 lanes serve one observation;
//   * the Zhang-Stephens grid (psis.py:163-208) is three grid points b_j per lane; the factors
//     prod_i (1 - b_j y_i) are taken four y at a time as a quartic in the grid coordinate whose five
//     coefficients each lane builds for its own quads and every lane of the row reads back from LDS
//     (same-address reads: a broadcast);
//   * reductions are four DPP steps inside the row, never across rows.
// Same arithmetic as wave_back(): mantissa / exponent accumulators instead of sums of log1p, table-driven
// exp / log, loo_i from the algebraic shortcut.  A grid point with |b_j y_n| < 2^-14 (the product would lose
// the digits of b_j y) takes its sum of log1p from the power sums of the tail instead.  Observations with an
// unusual m_est, factors beyond 2^+-30, sigma <= 0, a tiny khat, a cancelling total or a non-finite result
// are appended to the device list for the general kernel, like the observations the wave kernel declined.
#pragma once

#include "pla_wave.h"

namespace pla {

constexpr int kFitWaves = 4;       // waves per workgroup (they share the tables); 2 for the long-tail instantiations (LDS)
constexpr int kFitGrid = 48;       // three grid points per lane: m_est = 30 + isqrt(n) <= 46 for n <= 256
constexpr int kFitGridBig = 64;    // four per lane: tails up to 448 values (m_est <= 51), S = 20 000 at reff = 1
constexpr int kFitCoefStride = 6;  // (VALU grid pass) doubles per quad in LDS, five used: 48 bytes keep the 16-byte alignment
#ifndef PLA_FIT_MFMA
#define PLA_FIT_MFMA 1  // 1: the quartics of the grid pass on the matrix cores (v_mfma_f64_16x16x4_f64); 0: Horner on the VALU
#endif
#ifndef PLA_FIT_PIN
#define PLA_FIT_PIN 1  // pairs of tail elements of the smoothing pass the scheduler may interleave
#endif
#ifndef PLA_FIT_MIN_WAVES
#define PLA_FIT_MIN_WAVES 2  // waves per SIMD the fit kernel is compiled for
#endif
#ifndef PLA_FIT_SORT_ROUNDS
#define PLA_FIT_SORT_ROUNDS 3  // rounds (two passes each) of the odd-even transposition sort before the first look at whether the rows are sorted
#endif
#ifndef PLA_STREAM_SLEEP
#define PLA_STREAM_SLEEP 32  // s_sleep argument between two looks at a chunk's flag (units of 64 cycles)
#endif
typedef double v4d __attribute__((ext_vector_type(4)));

struct FitParams {
  const double* ws_y;   // [n_obs][ws_stride] the tail's shifted log ratios x, grouped by selection bin (bins descending), padded with the cutoff; ws_stride = 64 NQ
  const double* ws_s;   // [n_obs][ws_sstride]: max raw, min raw, sum_all e^x, sum_all e^-x, xcut, n (-1: declined), -, -
  int ws_stride;
  int64_t n_obs;
  int n_draws;
  int tail_count;
  int mest_M;
  double log_S;
  double scale_value;
  const double* l1_table;  // [M] log1p(-(j+0.5)/M)
  const double* b_grid;    // [64] 1 - sqrt(m_est/(j+0.5)) for m_est = mest_M
  double* diag;
  double* loo_i;
  double* lppd_i;
  unsigned* slow_list;
  unsigned long long* counters;
  unsigned slow_base = 0;  // added to the row numbers written to slow_list
  int ws_sstride = 8;      // doubles per observation in ws_s
  // streamed pass (fit_rows_stream_kernel): chunks of kQueueChunk observations are taken from `take` in order, each once
  // `done[c]` is set by the wave kernel running beside this one; `fitted[c]` is set behind the chunk's outputs.  Waiting is
  // bounded in TIME (s_memrealtime, 100 MHz): when a chunk's flag has not come for `patience` ticks the producer's row queue
  // (`producer`) is looked at, and if that has not moved either -- the two kernels are not running side by side after all, e.g.
  // under a profiler that serialises them -- `gave_up` is set and the chunks not fitted are left to the plain fit kernel, which
  // the launcher always runs behind both (fit_rows_kernel with `fitted`: it only looks at chunks whose flag is clear, i.e. at
  // nothing in the usual case).  `patience_start` applies while the producer has not taken a single row yet.
  const unsigned* done = nullptr;
  unsigned* take = nullptr;
  unsigned* fitted = nullptr;
  unsigned* gave_up = nullptr;
  const unsigned* producer = nullptr;
  unsigned patience = 0, patience_start = 0;
  unsigned long long* gave_up_total = nullptr;  // engine statistics: passes in which the streamed fit gave up (plain fit kernel, leftovers)
  // WEIGHTS mode (psislw of rows longer than the registers: selection kernel -> this kernel -> lw_output_kernel, pla_lwout.h):
  // this kernel writes, per observation, the tail's x back to ws_y SORTED (descending rank p at [p]) and into the scalars
  // log(sum of the smoothed weights) at [2], k-hat at [3], sigma / k-hat at [6], e_cut - sigma / k-hat at [7] -- what the smoothed
  // weight of a rank is evaluated from -- and the number of tail draws to patch at [5] (0: nothing was smoothed; -1: on the
  // list for the general kernel); k-hat goes to `diag` as in LOO mode
  bool lw_mode = false;
};

// 16 bytes of the hand-over.  STREAM: an agent-scope (sc1) load straight from memory -- the bytes were written, by the kernel
// running beside this one, after this kernel started (no kernel boundary in between: see FastParams::done)
template <bool STREAM>
__device__ __forceinline__ double2 ws_load2(const double* base, const int64_t elem) {
  if constexpr (STREAM) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    // (offsets are unsigned 32-bit: a block of 2^20 observations at the widest hand-over stride, 256 doubles, ends at 2^31 bytes)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, (int)0xfffffff0u, 0x00020000);
    const v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(unsigned)((uint64_t)elem * 8u), 0, 16);  // aux 16 = sc1
    return make_double2(__hiloint2double(t[1], t[0]), __hiloint2double(t[3], t[2]));
  } else {
    return *reinterpret_cast<const double2*>(base + elem);
  }
}

// reduction over the 16 lanes of a DPP row; every lane of the row ends up with the (bitwise identical) result
template <class F>
__device__ __forceinline__ double row_all(double v, F op) {
  v = op(v, dpp_mov_u<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_mov_u<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = op(v, dpp_mov_u<0x141, 0xF>(v));  // row_half_mirror
  v = op(v, dpp_mov_u<0x140, 0xF>(v));  // row_mirror
  return v;
}
__device__ __forceinline__ int row_all_add(int v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false);
  v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, false);
  return v;
}

// NQ: 64-value blocks of the tail (4 NQ values per lane); G: grid points per lane (16 G >= m_est); W: waves per workgroup
// DYN: the coefficient scratch is the kernel's dynamic LDS (fit_coef_bytes).  The compiler then cannot see that LDS caps the
// kernel at three waves per SIMD and holds the register budget its launch bounds ask for instead of relaxing it to that cap.
template <int NQ, int W>
constexpr size_t fit_coef_bytes() { return (size_t)W * 5 * 4 * (16 * NQ) * sizeof(double); }
template <int NQ, int G, int W, bool DYN = false, bool STREAM = false>
__device__ __forceinline__ void fit_rows_body(const FitParams& Q) {
  static_assert(!STREAM || W * 4 == kQueueChunk, "streamed pass: one chunk of the wave kernel per workgroup and trip");
  if constexpr (!STREAM) {
    // behind a streamed pass, which fitted everything unless it gave up waiting: nothing to do, not even the tables
    if (Q.fitted && Q.gave_up && *Q.gave_up == 0u) return;
    if (Q.fitted && Q.gave_up_total && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(Q.gave_up_total, 1ull);
  }
  __shared__ int s_chunk[2];
  constexpr int kFitWaves = W;  // (shadows the default: everything below is per instantiation)
  __shared__ __attribute__((aligned(16))) double tab[2 * kTabN];
  __shared__ __attribute__((aligned(16))) double lt[2 * kLogTabN];
  __shared__ __attribute__((aligned(16))) double l1s[64 * NQ];
#if PLA_FIT_MFMA
  constexpr int QN = 16 * NQ;  // quads per observation
  // Long tails (NQ >= 5: the two-wave instantiations behind the chunked kernel) write and consume the quartic coefficients in
  // TWO parts of KH 64-value blocks each: the scratch of a wave is then what the sort needs (4 x 64 NQ doubles) instead of the
  // 5 x 4 x 16 NQ of all coefficients at once -- 14.3 instead of 17.9 KB at NQ = 7, four workgroups per CU instead of three
  // (eight waves: what the registers allow anyway)
  constexpr int kParts = (NQ >= 5 && !DYN) ? 2 : 1;
  constexpr int KH = (NQ + kParts - 1) / kParts;  // 64-value blocks per part
  constexpr int QH = 16 * KH;                      // quads per observation and part
  constexpr int kWaveScratch = (5 * 4 * QH > 4 * 64 * NQ) ? 5 * 4 * QH : 4 * 64 * NQ;  // doubles (the sort's blocked copy lies here too)
  extern __shared__ __attribute__((aligned(16))) double coef_dyn[];
  __shared__ __attribute__((aligned(16))) double coef_static[DYN ? 2 : kFitWaves * kWaveScratch];  // per wave [coefficient k][observation][quad]
  double* const coef = DYN ? coef_dyn : coef_static;
#else
  __shared__ __attribute__((aligned(16))) double coef[kFitWaves * 4 * 16 * NQ * kFitCoefStride];
#endif
  const int tid = threadIdx.x;
  const int M = Q.tail_count, S = Q.n_draws, mestM = Q.mest_M;
  for (int j = tid; j < kTabN; j += kWave * kFitWaves) exp_table_entry(tab, j);
  for (int j = tid; j < kLogTabN; j += kWave * kFitWaves) log_table_entry(lt, j);
  if constexpr (kFitSorts) {
    // the register layout of the sorted tail: lane t holds the DESCENDING ranks p = K t + i (K = 4 NQ), i.e. the reference's
    // ascending index j = n - 1 - p.  The quantile table is stored reversed (for the usual n == M) and in the order the lanes
    // read it: entry (i / 2) * 32 + 2 t + (i & 1) belongs to p = K t + i  (conflict-free 16-byte reads).
    for (int idx = tid; idx < 64 * NQ; idx += kWave * kFitWaves) {
      const int p = (4 * NQ) * ((idx & 31) >> 1) + 2 * (idx >> 5) + (idx & 1);
      const int j = M - 1 - p;
      l1s[idx] = Q.l1_table[j < 0 ? 0 : j];
    }
  } else {
    for (int j = tid; j < 64 * NQ; j += kWave * kFitWaves) l1s[j] = Q.l1_table[j < M ? j : M - 1];
  }
  // (first and last grid point: read from LDS where they are used -- held in scalar registers across the group loop they are
  // four of the registers the streamed kernel does not have)
  __shared__ double g_ends[2];
  if (tid == 0) {
    g_ends[0] = Q.b_grid[0];
    g_ends[1] = Q.b_grid[mestM - 1];
  }
  __syncthreads();
  const double INF = pinf();
  const auto op_sum = [](double a, double b) { return a + b; };
  const auto op_mul = [](double a, double b) { return a * b; };
  const auto op_max = [](double a, double b) { return vmax_nc<false>(a, b); };
  const int lane = wave_lane(), t = lane & 15, rho = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
#if PLA_FIT_MFMA
  double* ck = coef + (size_t)wv * kWaveScratch;
  double* ys_base = ck;  // (sorting scratch: 4 x 64 NQ doubles of this wave's coefficient area, which is written afterwards)
  // One MFMA evaluates 4 quads x 4 observations x 16 grid points: D[row][col] = C0 + sum_k A[row][k] B[k][col] with
  // tile row = observation + 4 * quad (so the four results a lane receives, rows rho + 4 i at column t, are four
  // quads of the lane's OWN observation at its own grid point), A[row][k] = C_(k+1) of that quad and
  // B[k][col] = g_col^(k+1).  Operand lane (t, rho) supplies A[row t][k rho] and B[k rho][col t].
  const double* ckA = ck + ((rho + 1) * 4 + (t & 3)) * QH + (t >> 2);
  const double* ckC = ck + rho * QH;
  double gp[G];
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const int j = t + 16 * c;
    const double gj = Q.b_grid[j < mestM ? j : 0], g2 = gj * gj;
    gp[c] = rho == 0 ? gj : (rho == 1 ? g2 : (rho == 2 ? g2 * gj : g2 * g2));
  }
#else
  double* cf_row = coef + (size_t)((wv * 4 + rho) * 16 * NQ) * kFitCoefStride;
  double* ys_base = coef + (size_t)(wv * 4 * 16 * NQ) * kFitCoefStride;
#endif
  (void)ys_base;
  // grid coordinates of this lane: j = t, t + 16, t + 32 (, t + 48)
  double g[G];
  bool gact[G];
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const int j = t + 16 * c;
    gact[c] = j < mestM;
    g[c] = Q.b_grid[gact[c] ? j : 0];
  }
  const int64_t ngroups = (Q.n_obs + 3) >> 2;
  const int t_lane = t;
  const int64_t nchunks = (Q.n_obs + kQueueChunk - 1) / kQueueChunk;
  // one group of four observations, one per 16-lane row of this wave
  const auto fit_group = [&](const int64_t grp) __attribute__((always_inline)) {
    // (opaque per group: the ranks, table offsets and LDS addresses derived from the lane number are then computed where they
    // are used instead of being hoisted above the loop, where a dozen of them sit on the register budget of the slim variant)
    int t = t_lane;
    if constexpr (DYN) asm volatile("" : "+v"(t));
    PLA_PHASE(20);
    // (weights mode: asked of the kernel's argument block once per group, behind an opaque pointer -- as a loop invariant its
    // tests were hoisted above the group loop into scalar registers the kernel does not have: four spilled into vector lanes)
    bool lwm = false;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!STREAM) {
      typedef const __attribute__((address_space(4))) FitParams* ArgPtrG;
      ArgPtrG qa = (ArgPtrG)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(qa));
      lwm = qa->lw_mode;
    }
#endif
    const int64_t r0 = grp * 4 + rho;
    const bool inrange = r0 < Q.n_obs;
    const int64_t r = inrange ? r0 : Q.n_obs - 1;
    const int64_t sc = r * Q.ws_sstride;
    const double2 sc01 = ws_load2<STREAM>(Q.ws_s, sc), sc23 = ws_load2<STREAM>(Q.ws_s, sc + 2), sc45 = ws_load2<STREAM>(Q.ws_s, sc + 4);
    const double m = sc01.x, mn = sc01.y, s1 = sc23.x, s2 = sc23.y, xcut = sc45.x;
    const int nraw = (int)sc45.y;
    const bool handled = inrange && nraw >= 0;  // (else: past the end, or declined by the wave kernel and already listed)
    const int n = nraw < 0 ? 0 : (nraw > M ? M : nraw);
    const bool fit = n > 4;  // psis.py:139: shorter tails are neither fitted nor smoothed (their x was not written)
    const double e_cut = exp_tab(xcut, tab);  // (hand-over: shifted log ratios; the exponentials are taken here)
    const int64_t y0 = r * (int64_t)Q.ws_stride;
    const double* y = Q.ws_y + y0;
    double2 yv[2 * NQ];  // element j = 32 i + 2 t + {0, 1}
#pragma unroll
    for (int i = 0; i < 2 * NQ; ++i) yv[i] = ws_load2<STREAM>(Q.ws_y, y0 + 32 * i + 2 * t);
    const int iq = ((n + 2) >> 2) - 1;
    double yq, yn;
    if constexpr (kFitSorts) {
      constexpr int K = 4 * NQ;
    PLA_PHASE(21);
      // interleaved (as loaded) -> blocked, through this wave's LDS scratch (the coefficient area: written later)
      double* ys = ys_base + rho * (64 * NQ);
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) *reinterpret_cast<double2*>(ys + 32 * i + 2 * t) = yv[i];
      wave_sync();
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) {
        yv[i] = *reinterpret_cast<const double2*>(ys + K * t + 2 * i);
        if (!fit) yv[i] = make_double2(0.0, 0.0);  // (nothing was handed over: whatever the buffer held must not reach the sort)
      }
    PLA_PHASE(22);
      // descending odd-even transposition sort of the row's 16 K values: yb(i) = rank K t + i
      const auto yb = [&](int i) -> double& { return (i & 1) ? yv[i >> 1].y : yv[i >> 1].x; };
      const auto ce = [&](double& hi, double& lo) {
        const double a = vmax_nc<false>(hi, lo), b = vmin_nc(hi, lo);
        hi = a;
        lo = b;
      };
      const auto sort_round = [&]() {
#pragma unroll
        for (int i = 0; i < K; i += 2) ce(yb(i), yb(i + 1));
#pragma unroll
        for (int i = 1; i < K - 1; i += 2) ce(yb(i), yb(i + 1));
        // the pair that straddles two lanes: rank K t + K - 1 (this lane) and K (t + 1) (the next lane's first)
        const double right0 = dpp_mov<0x101, 0xF>(yb(0), -INF);     // row_shl:1; lane 15 of the row: nothing to its right
        const double leftk = dpp_mov<0x111, 0xF>(yb(K - 1), INF);   // row_shr:1; lane 0: nothing to its left
        yb(K - 1) = vmax_nc<false>(yb(K - 1), right0);
        yb(0) = vmin_nc(yb(0), leftk);
      };
      const auto unsorted = [&]() {
        bool u = yb(K - 1) < dpp_mov<0x101, 0xF>(yb(0), -INF);
#pragma unroll
        for (int i = 0; i < K - 1; ++i) u = u || (yb(i) < yb(i + 1));
        return __ballot(u) != 0ull;
      };
      // bins of the selection histogram hold <= 6 values in all but a few per cent of the rows: three rounds (six passes)
      // finish those; the check keeps going for the rest (16 K passes sort ANY order, the cap is that bound)
#pragma unroll 1
      for (int it = 0; it < PLA_FIT_SORT_ROUNDS; ++it) sort_round();
#pragma unroll 1
      for (int it = 0; it < 8 * K && unsorted(); ++it) sort_round();
    PLA_PHASE(23);
      if constexpr (!STREAM) {
        if (lwm && handled && fit) {  // weights mode: the output pass ranks the row's tail draws against this (pla_lwout.h)
          double* ysorted = const_cast<double*>(Q.ws_y) + y0;
#pragma unroll
          for (int i = 0; i < 2 * NQ; ++i) *reinterpret_cast<double2*>(ysorted + K * t + 2 * i) = yv[i];
        }
      }
      // y = e^x - e^xcut (psis.py:147) of the sorted tail; ranks from n on are padding (x = xcut): y = 0 exactly
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) yv[i] = make_double2(exp_tab(yv[i].x, tab) - e_cut, exp_tab(yv[i].y, tab) - e_cut);
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) *reinterpret_cast<double2*>(ys + K * t + 2 * i) = yv[i];
      wave_sync();
      const int pq = n - 1 - (iq > 0 ? iq : 0);
      yq = ys[pq > 0 ? pq : 0];  // psis.py:187: ascending element int(n/4 + 0.5) - 1
      yn = ys[0];                // psis.py:188: the largest
    } else {
      yq = y[iq > 0 ? iq : 0];
      yn = y[n > 0 ? n - 1 : 0];
    }
    PLA_PHASE(24);
    const double R = m - mn;
    const double nn = (double)n, rn = recip_fast(nn);
    bool bad = (30 + isqrt_i(n)) != mestM;  // (ties shortened the tail a lot: the host grid table does not apply)
    // b_j = g_j / (3 yq) + 1 / yn (psis.py:186-188), kept as the two per-observation scalars
    const double cb = recip_fast(3.0 * yq), db = recip_fast(yn);
    {
      const double g_first = g_ends[0], g_last = g_ends[1];
      const double fbig = fma(-fma(g_first, cb, db), yn, 1.0), fsmall = fma(-fma(g_last, cb, db), yn, 1.0);
      if (!((fbig < 0x1p30) && (fsmall > 0x1p-30))) bad = true;
    }
    PLA_PHASE(25);
    // ---- quartic coefficients of this lane's quads --------------------------------------------------------
    // 1 - b_j y = (1 - y/yn) - g_j y/(3 yq) = u + g_j t with u >= 0, t <= 0 and every g_j < 0: the product over
    // four y is a quartic in g_j whose terms are all non-negative (no cancellation)
    double pm[G];
    int pe[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      pm[c] = 1.0;
      pe[c] = 0;
    }
#if PLA_FIT_MFMA
#pragma unroll
    for (int part = 0; part < kParts; ++part) {
      constexpr int kLastBlocks = NQ - (kParts - 1) * KH;  // blocks of the last part
      const int k0 = part * KH, nk = (part == kParts - 1) ? kLastBlocks : KH;
      if (part > 0) wave_sync();  // (the part before has been read)
#pragma unroll
      for (int kk = 0; kk < KH; ++kk) {
        if (kk < nk) {
          const int k = k0 + kk;
          const double y0 = yv[2 * k].x, y1 = yv[2 * k].y, y2 = yv[2 * k + 1].x, y3 = yv[2 * k + 1].y;
          const double u0 = fma(-db, y0, 1.0), u1 = fma(-db, y1, 1.0), u2 = fma(-db, y2, 1.0), u3 = fma(-db, y3, 1.0);
          const double t0 = -cb * y0, t1 = -cb * y1, t2 = -cb * y2, t3 = -cb * y3;
          const double A0 = u0 * u1, A1 = fma(u0, t1, u1 * t0), A2 = t0 * t1;
          const double B0 = u2 * u3, B1 = fma(u2, t3, u3 * t2), B2 = t2 * t3;
          double* o = ck + rho * QH + (kk * 16 + t);
          o[0] = A0 * B0;
          o[4 * QH] = fma(A0, B1, A1 * B0);
          o[8 * QH] = fma(A0, B2, fma(A1, B1, A2 * B0));
          o[12 * QH] = fma(A1, B2, A2 * B1);
          o[16 * QH] = A2 * B2;
        }
      }
      wave_sync();
      PLA_PHASE(26);
      // ---- grid pass: three running products per lane over the 16 NQ quads of the observation -------------------
#pragma unroll 1
      for (int q4 = 0; q4 < 16 * nk; q4 += 8) {
#pragma unroll
        for (int u = 0; u < 8; u += 4) {
          const double a = ckA[q4 + u];
          const double2 c01 = *reinterpret_cast<const double2*>(ckC + q4 + u), c23 = *reinterpret_cast<const double2*>(ckC + q4 + u + 2);
          const v4d cin = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
          for (int c = 0; c < G; ++c) {
            const v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, gp[c], cin, 0, 0, 0);
            pm[c] *= (d[0] * d[1]) * (d[2] * d[3]);
          }
        }
        // factors within 2^+-120 per quad: eight fit between renormalisations
#pragma unroll
        for (int c = 0; c < G; ++c) {
          pe[c] += __builtin_amdgcn_frexp_exp(pm[c]);
          pm[c] = __builtin_amdgcn_frexp_mant(pm[c]);
        }
      }
    }
#else
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const double y0 = yv[2 * k].x, y1 = yv[2 * k].y, y2 = yv[2 * k + 1].x, y3 = yv[2 * k + 1].y;
      const double u0 = fma(-db, y0, 1.0), u1 = fma(-db, y1, 1.0), u2 = fma(-db, y2, 1.0), u3 = fma(-db, y3, 1.0);
      const double t0 = -cb * y0, t1 = -cb * y1, t2 = -cb * y2, t3 = -cb * y3;
      const double A0 = u0 * u1, A1 = fma(u0, t1, u1 * t0), A2 = t0 * t1;
      const double B0 = u2 * u3, B1 = fma(u2, t3, u3 * t2), B2 = t2 * t3;
      double* o = cf_row + (k * 16 + t) * kFitCoefStride;
      *reinterpret_cast<double2*>(o) = make_double2(A0 * B0, fma(A0, B1, A1 * B0));
      *reinterpret_cast<double2*>(o + 2) = make_double2(fma(A0, B2, fma(A1, B1, A2 * B0)), fma(A1, B2, A2 * B1));
      o[4] = A2 * B2;
    }
    wave_sync();
    PLA_PHASE(26);
#pragma unroll 1
    for (int q8 = 0; q8 < 16 * NQ; q8 += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double* cq = cf_row + (q8 + u) * kFitCoefStride;
        const double2 c01 = *reinterpret_cast<const double2*>(cq), c23 = *reinterpret_cast<const double2*>(cq + 2);
        const double c4 = cq[4];
#pragma unroll
        for (int c = 0; c < G; ++c) pm[c] *= fma(g[c], fma(g[c], fma(g[c], fma(g[c], c4, c23.y), c23.x), c01.y), c01.x);
      }
      // factors within 2^+-120 per quad: eight fit between renormalisations
#pragma unroll
      for (int c = 0; c < G; ++c) {
        pe[c] += __builtin_amdgcn_frexp_exp(pm[c]);
        pm[c] = __builtin_amdgcn_frexp_mant(pm[c]);
      }
    }
#endif
    wave_sync();  // (the next group's coefficients are written after these reads)
    PLA_PHASE(27);
    // ---- profile likelihood, softmax weights, posterior mean of b (psis.py:190-201) --------------------------
    double ls[G], bb[G], lp[G];
    bool tiny[G];
#pragma unroll
    for (int c = 0; c < G; ++c) {
      bb[c] = fma(g[c], cb, db);
      tiny[c] = gact[c] && fabs(bb[c] * yn) < 0x1p-14;  // 1 - b y rounds away the digits of b y
      lp[c] = log_tab(pm[c], lt) + (double)pe[c] * kLn2;
    }
    bool any_tiny = false;
#pragma unroll
    for (int c = 0; c < G; ++c) any_tiny = any_tiny || tiny[c];
    if (__ballot(any_tiny) != 0ull) {
      // (one observation in ~2000) sum_i log1p(-b y_i) = -(b p1 + b^2 p2/2 + b^3 p3/3 + b^4 p4/4) + O((b y)^5) from the
      // power sums p_k of the tail: to 2^-56 relative for |b y| < 2^-14
      double p1 = 0.0, p2 = 0.0, p3 = 0.0, p4 = 0.0;
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const double yj = e ? yv[i].y : yv[i].x, y2 = yj * yj;
          p1 += yj;
          p2 += y2;
          p3 = fma(y2, yj, p3);
          p4 = fma(y2, y2, p4);
        }
      }
      p1 = row_all(p1, op_sum);
      p2 = row_all(p2, op_sum);
      p3 = row_all(p3, op_sum);
      p4 = row_all(p4, op_sum);
#pragma unroll
      for (int c = 0; c < G; ++c) {
        const double b = bb[c];
        if (tiny[c]) lp[c] = -b * fma(b, fma(b, fma(b, 0.25 * p4, p3 * (1.0 / 3.0)), 0.5 * p2), p1);
      }
    }
#pragma unroll
    for (int c = 0; c < G; ++c) {
      const double b = bb[c];
      const double kj = lp[c] * rn;                                                               // psis.py:190
      const double l = nn * (log_tab(gact[c] ? -div_fast(b, kj) : 1.0, lt) - kj - 1.0);           // psis.py:191
      if (gact[c] && l != l) bad = true;  // NaN anywhere: every weight is NaN in the reference; the general kernel does that
      ls[c] = gact[c] ? l : -INF;
    }
    double lmx = ls[0];
#pragma unroll
    for (int c = 1; c < G; ++c) lmx = vmax_nc<false>(lmx, ls[c]);
    const double lmax = row_all(lmx, op_max);
    if (!(fabs(lmax) < INF)) bad = true;
    double w[G];
#pragma unroll
    for (int c = 0; c < G; ++c) w[c] = gact[c] ? exp_neg(ls[c] - lmax, tab) : 0.0;                // psis.py:192
    double wsum = w[0];
#pragma unroll
    for (int c = 1; c < G; ++c) wsum += w[c];
    const double se = row_all(wsum, op_sum);
    // the weights stay unnormalised (b_post is a ratio): w/se >= 10 eps  <=>  w >= 10 eps se   (psis.py:194-197)
    double swl = 0.0, bwl = 0.0;
#pragma unroll
    for (int c = 0; c < G; ++c) {
      const bool keep = w[c] >= (10.0 * kEps) * se;
      swl += keep ? w[c] : 0.0;
      bwl += keep ? bb[c] * w[c] : 0.0;
    }
    const double sw = row_all(swl, op_sum), bw = row_all(bwl, op_sum);
    const double b_post = (sw > 0.0) ? div_fast(bw, sw) : 0.0;                                    // psis.py:198,201
    PLA_PHASE(28);
    // ---- k_post = mean log1p(-b_post y) (psis.py:203) from the lane's own quads --------------------------------
    double km = 1.0;
    {
      const double nb = -b_post;
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const double y0 = yv[2 * k].x, y1 = yv[2 * k].y, y2 = yv[2 * k + 1].x, y3 = yv[2 * k + 1].y;
        const double s01 = y0 + y1, p01 = y0 * y1, s23 = y2 + y3, p23 = y2 * y3;
        const double e1 = s01 + s23, e2 = fma(s01, s23, p01 + p23), e3 = fma(p01, s23, p23 * s01), e4 = p01 * p23;
        km *= fma(nb, fma(nb, fma(nb, fma(nb, e4, e3), e2), e1), 1.0);
      }
    }
    const int ke = row_all_add(__builtin_amdgcn_frexp_exp(km));
    const double kmr = row_all(__builtin_amdgcn_frexp_mant(km), op_mul);  // >= 2^-16
    const double k_post = (log_tab(kmr, lt) + (double)ke * kLn2) * rn;
    const double sigma = -k_post / b_post;                                                        // psis.py:205
    const double khat = (nn * k_post + 5.0) / (nn + 10.0);                                        // psis.py:206
    PLA_PHASE(29);
    // ---- smoothed tail: sums of w'_j - e_j and w'_j / e_j (psis.py:150-158, 211-231) ---------------------------
    const bool smoothed = fit && isfinite(khat);
    // w_j = sigma/k (e^{z_j} - 1) + e_cut with z_j = -k log1p(-p_j); e^z - 1 by subtraction is accurate to 1e-16
    // ABSOLUTE, which is all the sum w_j + e_cut can see.  sigma <= 0 (NaN quantiles, psis.py:214-215) and
    // |k| < eps (psis.py:218) are left to the general kernel.
    if (smoothed && (!(sigma > 0.0) || fabs(khat) < kEps)) bad = true;
    const double coef_s = sigma / khat, off = e_cut - coef_s;
    double acc_t = 0.0, acc_r = 0.0;
    const auto smooth = [&](int j, double l1, double yj) {
      const double ez = exp_tab(fmin(-khat * l1, 700.0), tab);
      const double wj = fmin(fma(ez, coef_s, off), 1.0);  // exp(log(q + e_cut)) clipped at 0 (psis.py:155,157)
      const double ej = yj + e_cut;
      acc_t += (j < n) ? wj - ej : 0.0;
      acc_r += (j < n) ? div_fast(wj, ej) : 0.0;
    };
    // (the sums are pinned every few elements: left alone, the scheduler starts every element at once and spills)
    const auto pin = [&]() { asm volatile("" : "+v"(acc_t), "+v"(acc_r)); };
    if constexpr (kFitSorts) {
      constexpr int K = 4 * NQ;
      const bool usual = __ballot(fit && n != M) == 0ull;
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) {
        const int p = K * t + 2 * i;  // descending rank of yv[i].x; the reference's ascending index is n - 1 - p (psis.py:146,153)
        double2 l1 = *reinterpret_cast<const double2*>(l1s + 32 * i + 2 * t);
        if (!usual) {  // ties shortened some tail of this wave: p_j = (j + 0.5)/n with the observation's own n
          asm volatile("");
          if (n != M) l1 = make_double2(log_fast(1.0 - ((double)(n - 1 - p) + 0.5) * rn), log_fast(1.0 - ((double)(n - 2 - p) + 0.5) * rn));
        }
        smooth(p, l1.x, yv[i].x);
        smooth(p + 1, l1.y, yv[i].y);
        if ((i % PLA_FIT_PIN) == PLA_FIT_PIN - 1) pin();
      }
    } else if (__ballot(fit && n != M) == 0ull) {
      // the usual case, straight-line: log1p(-p_j) from the host table (psis.py:153)
#pragma unroll
      for (int i = 0; i < 2 * NQ; ++i) {
        const int j = 32 * i + 2 * t;
        const double2 l1 = *reinterpret_cast<const double2*>(l1s + j);
        smooth(j, l1.x, yv[i].x);
        smooth(j + 1, l1.y, yv[i].y);
        if ((i % PLA_FIT_PIN) == PLA_FIT_PIN - 1) pin();
      }
    } else {
      // ties shortened some tail of this wave: p_j = (j + 0.5)/n with the observation's own n
#pragma unroll 1
      for (int i = 0; i < 4 * NQ; ++i) {
        const int j = 16 * i + t;
        smooth(j, n == M ? l1s[j] : log_fast(1.0 - ((double)j + 0.5) * rn), y[j]);
        pin();
      }
    }
    PLA_PHASE(30);
    const double at = row_all(acc_t, op_sum), ar = row_all(acc_r, op_sum);
    // ---- outputs (loo.py:289,319-337 through the shortcuts of DESIGN.md section 4) --------------------------
    const double total = smoothed ? s1 + at : s1;
    bool bad_out = !(total > kCancelGuard * s1);  // the tail cancels against the sum of all exponentials
    const double tail_ratio = smoothed ? (double)(S - n) + ar : (double)S;
    // (weights mode: lane 0's logarithm is log(total) itself (psis.py:158), the second sum is not looked at)
    const double lg = log_tab(t == 1 ? (lwm ? 1.0 : s2) : (lwm ? total : div_fast(tail_ratio, total)), lt);
    const double lg1 = dpp_mov_u<0xB1, 0xF>(lg);  // lane t = 0 receives lane 1's logarithm
    const double loo = lg - m;
    const double lppd = lwm ? 0.0 : (lg1 - R) + ((-mn) - Q.log_S);
    if (!(total > 1e-280) || !isfinite(loo) || !isfinite(lppd)) bad_out = true;
    const unsigned long long badm = __ballot(bad);
    const bool slow = (fit && ((badm >> (lane & 48)) & 0xFFFFull) != 0ull) || bad_out;
    // (streamed kernel: what only the end of a group needs is read from the kernel's argument block HERE -- scalar loads from
    // constant memory -- instead of living in scalar registers from the top of the kernel through every group: it has none to
    // spare, and scalars it spilled into vector lanes have come back wrong -- see wait_for below)
    typedef const __attribute__((address_space(4))) FitParams* ArgPtr;  // (constant address space: scalar loads)
    double *pd = Q.diag, *pl = Q.loo_i, *pp = Q.lppd_i;
    unsigned long long* pcount = Q.counters;
    unsigned* plist = Q.slow_list;
    unsigned sbase = Q.slow_base;
    double scale = Q.scale_value;
#if defined(__HIP_DEVICE_COMPILE__)
    {  // (both kernels: the plain one keeps them no better -- four of these tests sat in vector lanes once weights mode had added two scalars)
      ArgPtr qp = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(qp));
      pd = qp->diag; pl = qp->loo_i; pp = qp->lppd_i;
      pcount = qp->counters; plist = qp->slow_list; sbase = qp->slow_base; scale = qp->scale_value;
    }
#endif
    if (handled && t == 0) {
      if (slow) {
        const unsigned long long idx = atomicAdd(&pcount[0], 1ull);
        plist[idx] = (unsigned)r + sbase;
      } else {
        if (pd) pd[r] = fit ? khat : INF;
        if (pl) pl[r] = scale * loo;
        if (pp) pp[r] = lppd;
      }
      if (lwm) {  // for the output pass: log of the normaliser, tail draws to patch (-1: the general kernel writes this row)
        double* sc_out = const_cast<double*>(Q.ws_s) + sc;
        sc_out[2] = lg;
        sc_out[3] = khat;     // (the smoothed weight of a rank is min(coef (e^(-khat log1p(-p_j)) - 1) + e_cut, 1): psis.py:153-157)
        sc_out[6] = coef_s;
        sc_out[7] = off;
        sc_out[5] = slow ? -1.0 : (smoothed ? (double)n : 0.0);
      }
    }
  };
  if constexpr (STREAM) {
    // The chunks in order, each once the wave kernel has finished it.  One lane takes the next chunk number (a returning atomic:
    // issued before this chunk's arithmetic, used behind it), looks at its flag (agent-scope loads, a sleep in between) and
    // posts it in LDS; the workgroup barrier stands between that poll and every load of the chunk's bytes.
    // false: gave up (see FitParams::patience)
    // (the state of the wait lives in VECTOR registers -- one lane runs this -- by force: left to itself the compiler keeps the
    // clock, the queue position and the bounds in scalar registers, runs out of them and spills scalars into vector lanes
    // from inside this one-lane branch, which the other waves of the workgroup then read back without ever having written
    // them: groups of waves 1 and 3 went missing.  tests/test_kernel_resources.py holds this kernel to zero such spills.)
    const auto wait_for = [&](int c) -> bool {
      asm volatile("" : "+v"(c));
      // (done[c] counts the rows of chunk c that have been handed over: pla_fast.h, kQueueUnit)
      const int64_t left = Q.n_obs - (int64_t)c * kQueueChunk;
      unsigned need = left < kQueueChunk ? (unsigned)left : (unsigned)kQueueChunk;
      asm volatile("" : "+v"(need));
      if (__hip_atomic_load(Q.done + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
      // (32 bits of the 100 MHz counter: differences are good for 42 s)
      unsigned mark = (unsigned)__builtin_amdgcn_s_memrealtime();
      unsigned seen = __hip_atomic_load(Q.producer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("" : "+v"(mark), "+v"(seen));
      for (;;) {
        __builtin_amdgcn_s_sleep(PLA_STREAM_SLEEP);
        if (__hip_atomic_load(Q.done + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
        asm volatile("" : "+v"(now));
        if (now - mark > (seen ? Q.patience : Q.patience_start)) {
          const unsigned q = __hip_atomic_load(Q.producer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (q == seen) {  // the producer's row queue stands still: it is not running beside this kernel
            __hip_atomic_store(Q.gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
          }
          seen = q;
          mark = now;
          asm volatile("" : "+v"(mark), "+v"(seen));
        }
      }
    };
    if (tid == 0) {
      int c = (int)atomicAdd(Q.take, 1u);
      if ((int64_t)c >= nchunks || !wait_for(c)) c = -1;
      s_chunk[0] = c;
    }
    int prev_chunk = -1;
    for (int trip = 0;; ++trip) {
      __syncthreads();  // (s_chunk[trip & 1] is posted; everybody is done with the chunk of the previous trip)
      const int c = s_chunk[trip & 1];
      int nxt = -1;
      if (tid == 0) {
        if (prev_chunk >= 0) Q.fitted[prev_chunk] = 1u;
        if (c >= 0) nxt = (int)atomicAdd(Q.take, 1u);
      }
      prev_chunk = c;
      if (c < 0) break;
      const int64_t grp = (int64_t)c * (kQueueChunk / 4) + wv;
      if (grp < ngroups) fit_group(grp);  // (the last chunk may be short)
      if (tid == 0) {
        if ((int64_t)nxt >= nchunks || !wait_for(nxt)) nxt = -1;
        s_chunk[(trip + 1) & 1] = nxt;
      }
    }
  } else {
    for (int64_t grp = (int64_t)blockIdx.x * kFitWaves + wv; grp < ngroups; grp += (int64_t)gridDim.x * kFitWaves) {
      if (Q.fitted) {  // behind a streamed pass: only what that pass left (nothing, unless it gave up waiting)
        if (Q.fitted[grp / (kQueueChunk / 4)] != 0u) continue;
      }
      fit_group(grp);
    }
  }
}

template <int NQ, int G = 3, int W = kFitWaves>
__global__ __launch_bounds__(kWave * W, PLA_FIT_MIN_WAVES) void fit_rows_kernel(FitParams Q) {
  fit_rows_body<NQ, G, W>(Q);
}

// The same fit for the streamed pass (pla_kernels.hip, launch_wave): it runs BESIDE the wave kernel, in what two workgroups of
// that kernel leave free on a CU -- 512 - 2 x 192 = 128 vector registers per SIMD and 160 KB - 2 x 58 KB of LDS -- so it is
// compiled for at most 128 registers (four waves per workgroup, 39.5 KB of LDS: one workgroup per CU beside the wave kernel's
// two).  fit_rows_stream_kernel takes the chunks as the wave kernel finishes them.
#ifndef PLA_FIT_SLIM_WAVES
#define PLA_FIT_SLIM_WAVES 4  // waves per SIMD the register budget is derived from: 4 -> 128 registers (5 -> 96: spills)
#endif
template <int NQ, int G = 3>
__global__ __launch_bounds__(kWave * 4, PLA_FIT_SLIM_WAVES) void fit_rows_stream_kernel(FitParams Q) {
  fit_rows_body<NQ, G, 4, true, true>(Q);
}

}  // namespace pla
