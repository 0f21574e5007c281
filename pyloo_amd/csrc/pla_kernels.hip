// HIP kernels of the PSIS-LOO engine for gfx950 (MI355X).  One workgroup per observation.
//
// What one workgroup does for its row of S draws (reference: pyloo psis.py:114-160,
// utils.py:305-359, loo.py:289-337; the per-observation Python loop of utils.py:171-175 is the
// grid):
//   1. max / min / special-value scan                       (psis.py:134)
//   2. exact selection of the (M+1)-th largest shifted log ratio     (psis.py:135-136)
//   3. tail (strictly above the cutoff) -> LDS, sorted ascending      (psis.py:139-146)
//   4. generalised-Pareto fit on exp(tail)-exp(cutoff)                (psis.py:147-148,163-208)
//   5. replace the tail by GPD quantiles, clip at 0                   (psis.py:150-157,211-231)
//   6. log-sum-exp normaliser, loo_i and lppd_i                       (psis.py:158, loo.py:289-337)
// The weight matrix is never materialised in LOO mode: for draws outside the tail
// lw_s + ll_s == -max - LSE exactly in real arithmetic, so only the <= M tail terms need an exp.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdarg>
#include <cstdlib>

#include "pla_fast.h"
#include "pla_rows.h"
#include "pla_wave.h"
#include "pla_waic.h"
#include "pla_chunked.h"
#include "pla_is.h"
#include "pla_fit.h"
#include "pla_eloo.h"
#include "pla_col.h"
#include "pla_tile.h"

namespace pla {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
template <typename T, int BLOCK, int EPT, bool LW>
__global__ __launch_bounds__(BLOCK) void rows_kernel(RowsParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Smem sm = carve<BLOCK>(smem_raw, P.tail_cap);
  for (int64_t r = blockIdx.x; r < P.n_obs; r += gridDim.x) {
    const T* rp = reinterpret_cast<const T*>(P.in) + PLA_ROW_OFFSET(P, r);
    if constexpr (EPT == 0) {
      RowGlobal<T, !LW> row{rp, P.stride_draw, P.n_draws};
      process_row<RowGlobal<T, !LW>, T, BLOCK, LW>(row, P, sm, r);
    } else {
      RowRegs<T, !LW, BLOCK, EPT> row;
      row.load(rp, P.stride_draw, P.n_draws);
      process_row<RowRegs<T, !LW, BLOCK, EPT>, T, BLOCK, LW>(row, P, sm, r);
    }
  }
}

// rows the fast kernel handed over: same general pipeline, row indices from the device list
template <typename T, int BLOCK, bool LW>
__global__ __launch_bounds__(BLOCK) void slow_rows_kernel(RowsParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const Smem sm = carve<BLOCK>(smem_raw, P.tail_cap);
  const unsigned long long count = P.counters[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&P.counters[1], count);  // running total of this call
  for (unsigned long long i = blockIdx.x; i < count; i += gridDim.x) {
    const int64_t r = (int64_t)P.slow_list[i];
    const T* rp = reinterpret_cast<const T*>(P.in) + PLA_ROW_OFFSET(P, r);
    RowGlobal<T, !LW> row{rp, P.stride_draw, P.n_draws};
    process_row<RowGlobal<T, !LW>, T, BLOCK, LW>(row, P, sm, r);
  }
}

// loo.py:326-342 + 292-293 in two deterministic launches.  Stage 1: every workgroup reduces its own
// contiguous chunk to (n, sum, M2 about the chunk mean, ...); np.var is a two-pass formula as well
// and the chunk is L2-resident for the second pass.  Stage 2: one wave merges the chunk moments with
// the pairwise update of Chan, Golub & LeVeque (no cancellation) in a fixed order.
constexpr int kRedBlock = 256;
constexpr int kRedChunks = 1024;
constexpr int kRedSlots = 8;  // n, sum loo, M2, sum lppd, #high, #non-finite, min diag, unused

__global__ __launch_bounds__(kRedBlock) void reduce_stage1(ReduceParams P, double* part) {
  __shared__ double red[16];
  const int tid = threadIdx.x;
  const int64_t per = (P.n_obs + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = (lo + per < P.n_obs) ? lo + per : P.n_obs;
  double s_loo = 0.0, s_lppd = 0.0, n_high = 0.0, n_bad = 0.0, dmin = pinf();
  for (int64_t i = lo + tid; i < hi; i += kRedBlock) {
    if (P.loo_i) s_loo += P.loo_i[i];
    if (P.lppd_i) s_lppd += P.lppd_i[i];
    if (P.diag) {
      const double d = P.diag[i];
      if (d > P.good_k) n_high += 1.0;
      if (!isfinite(d)) n_bad += 1.0;
      dmin = fmin(dmin, d);
    }
  }
  s_loo = block_reduce<OpSum, kRedBlock>(s_loo, red);
  s_lppd = block_reduce<OpSum, kRedBlock>(s_lppd, red);
  n_high = block_reduce<OpSum, kRedBlock>(n_high, red);
  n_bad = block_reduce<OpSum, kRedBlock>(n_bad, red);
  dmin = block_reduce<OpMin, kRedBlock>(dmin, red);
  const double cnt = (double)(hi > lo ? hi - lo : 0);
  const double mean = cnt > 0 ? s_loo / cnt : 0.0;
  double m2 = 0.0;
  if (P.loo_i)
    for (int64_t i = lo + tid; i < hi; i += kRedBlock) {
      const double d = P.loo_i[i] - mean;
      m2 += d * d;
    }
  m2 = block_reduce<OpSum, kRedBlock>(m2, red);
  if (tid == 0) {
    double* o = part + (size_t)blockIdx.x * kRedSlots;
    o[0] = cnt; o[1] = s_loo; o[2] = m2; o[3] = s_lppd; o[4] = n_high; o[5] = n_bad; o[6] = dmin; o[7] = 0.0;
  }
}

struct Moments {  // count, mean and M2 of loo_i over a set of observations + the plain sums
  double n, mean, m2, s_loo, s_lppd, n_high, n_bad, dmin;
};
__device__ __forceinline__ void merge_moments(Moments& a, const Moments& b) {  // Chan, Golub & LeVeque
  if (b.n == 0.0) return;
  if (a.n == 0.0) { a = b; return; }
  const double delta = b.mean - a.mean, tot = a.n + b.n;
  a.m2 += b.m2 + delta * delta * a.n * b.n / tot;
  a.mean += delta * b.n / tot;
  a.n = tot;
  a.s_loo += b.s_loo; a.s_lppd += b.s_lppd; a.n_high += b.n_high; a.n_bad += b.n_bad;
  a.dmin = fmin(a.dmin, b.dmin);
}

// One wave: lane l merges chunks l*per .. (l+1)*per-1 in order, then the 64 partial results are merged
// by a fixed shuffle tree.  The grouping depends only on the chunk count, so the result is
// reproducible run to run.
__global__ __launch_bounds__(kWave) void reduce_stage2(ReduceParams P, const double* part, int nchunks) {
  const int lane = threadIdx.x;
  const int per = (nchunks + kWave - 1) / kWave;
  Moments a{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, pinf()};
  for (int c = lane * per; c < (lane + 1) * per && c < nchunks; ++c) {
    const double* o = part + (size_t)c * kRedSlots;
    Moments b{o[0], o[0] > 0.0 ? o[1] / o[0] : 0.0, o[2], o[1], o[3], o[4], o[5], o[6]};
    merge_moments(a, b);
  }
  for (int off = 1; off < kWave; off <<= 1) {
    Moments b;
    b.n = __shfl_down(a.n, off); b.mean = __shfl_down(a.mean, off); b.m2 = __shfl_down(a.m2, off);
    b.s_loo = __shfl_down(a.s_loo, off); b.s_lppd = __shfl_down(a.s_lppd, off);
    b.n_high = __shfl_down(a.n_high, off); b.n_bad = __shfl_down(a.n_bad, off); b.dmin = __shfl_down(a.dmin, off);
    if ((lane & (2 * off - 1)) == 0 && lane + off < kWave) merge_moments(a, b);
  }
  if (lane != 0) return;
  const double s_loo = a.s_loo, m2 = a.m2, s_lppd = a.s_lppd, n_high = a.n_high, n_bad = a.n_bad, dmin = a.dmin;
  P.agg[PLA_AGG_N] = (double)P.n_obs;
  P.agg[PLA_AGG_SUM_LOO] = s_loo;
  P.agg[PLA_AGG_M2_LOO] = m2;
  P.agg[PLA_AGG_SUM_LPPD] = s_lppd;
  P.agg[PLA_AGG_N_HIGH] = n_high;
  P.agg[PLA_AGG_N_NONFINITE] = n_bad;
  P.agg[PLA_AGG_MIN_DIAG] = dmin;
  P.agg[PLA_AGG_N_SLOW] = P.counters ? (double)P.counters[0] : 0.0;  // caller passes &counters[1]
}

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* ll, int64_t n_obs, int64_t n_draws, int64_t row0,
                                                   uint64_t seed, double k_lo, double k_hi,
                                                   double heavy_lo, double heavy_hi) {
  const int64_t total = n_obs * n_draws;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t il = e / n_draws, s = e - il * n_draws;
    const int64_t i = il + row0;
    const double uk = u01_open(splitmix64(~seed ^ (uint64_t)i));
    double k = k_lo + (k_hi - k_lo) * uk;
    const int r10 = (int)(i % 10);
    if (heavy_hi > heavy_lo && (r10 == 0 || r10 == 3 || r10 == 6)) k = heavy_lo + (heavy_hi - heavy_lo) * uk;
    const double u = u01_open(splitmix64(seed ^ (uint64_t)(i * n_draws + s)));
    const double E = -log1p(-u);
    const double c = -1.0 - (double)(i % 7) * 0.25;
    ll[e] = (T)(-k * E + c);
  }
}

// Synthetic rows as MCMC delivers them (bench.py --rows chain_ar1): `chains` chains stacked chain-major along the draws (the
// (chain, draw) -> __sample__ stack of loo.py:189), every chain a stationary AR(1) sequence in the draw index -- z_t = rho z_{t-1}
// + sqrt(1 - rho^2) eps_t, standard normal marginals -- mapped to Exp(1) marginals E = -log(1 - Phi(z)) and on to
// ll = -k_i (E + o_ic / k_i ...) exactly as the iid generator does (same k_i, c_i), plus a per-chain offset o_ic ~ N(0, off_sd^2)
// of the chain's log-likelihoods.  One thread per (observation, chain): the recursion is sequential in t.
template <typename T>
__global__ __launch_bounds__(256) void fill_chains_kernel(T* ll, int64_t n_obs, int64_t n_draws, int chains, double rho,
                                                          double off_sd, int64_t row0, uint64_t seed, double k_lo, double k_hi) {
  const int64_t total = n_obs * chains;
  const int64_t per = n_draws / chains;
  const double sd = sqrt(1.0 - rho * rho);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t il = e / chains;
    const int c = (int)(e - il * chains);
    const int64_t i = il + row0;
    const double uk = u01_open(splitmix64(~seed ^ (uint64_t)i));
    const double k = k_lo + (k_hi - k_lo) * uk;
    const double ci = -1.0 - (double)(i % 7) * 0.25;
    const auto normal = [&](uint64_t ctr) {  // Box-Muller on two counter-based uniforms
      const double u1 = u01_open(splitmix64(seed ^ (2 * ctr))), u2 = u01_open(splitmix64(seed ^ (2 * ctr + 1)));
      return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    };
    const double off = off_sd * normal((uint64_t)(0x7000000000000000ull + (uint64_t)(i * chains + c)));
    const int64_t t0 = c * per, t1 = (c == chains - 1) ? n_draws : t0 + per;
    double z = normal((uint64_t)(i * n_draws + t0));
    T* row = ll + il * n_draws;
    for (int64_t t = t0; t < t1; ++t) {
      if (t > t0) z = rho * z + sd * normal((uint64_t)(i * n_draws + t));
      const double E = -log(0.5 * erfc(z * 0.7071067811865476));  // Exp(1) with the dependence of z
      row[t] = (T)(-k * E + ci + off);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
int max_tail_count() { return 8192; }

// what the last launch_rows() of this thread launched (pla_engine_last_kernels: benchmark records name what ran)
static thread_local char g_last_kernels[320] = "";
static void note_kernels(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernels, sizeof(g_last_kernels), fmt, ap);
  va_end(ap);
}
const char* last_rows_kernels() { return g_last_kernels; }
template <typename T> static const char* tname() { return sizeof(T) == 8 ? "double" : "float"; }

template <typename T, int BLOCK, int EPT, bool LW>
static hipError_t launch_one(const RowsParams& p, hipStream_t stream) {
  const size_t lds = smem_bytes(BLOCK, p.tail_cap);
  auto kern = rows_kernel<T, BLOCK, EPT, LW>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int64_t grid = p.n_obs;
  const int64_t cap = 256 * 64;  // >> 256 CUs; rows are strided over the grid
  if (grid > cap) grid = cap;
  note_kernels("rows_kernel<%s, %d> (general kernel: one workgroup per observation)", tname<T>(), BLOCK);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BLOCK), lds, stream, p);
  return hipGetLastError();
}

// Speculative threshold of the wave kernel: the kq-th smallest (roughly) of the 64 per-lane maxima
// over the first gsz register slots.  For exchangeable draws a draw lies below it with probability F,
// F^gsz = kq/64.  Pick (gsz, kq) so that ~2.2(M+1) draws lie above, with kq large enough for the order
// statistic to be stable.  Returns false when no setting fits (the general kernel takes the call).
// `cand_cap`: capacity of the LDS candidate list; only the first min(S, 4096) draws feed the maxima.
static int debug_flag(const char* name);
struct ThresholdCheck {  // FastParams::cr_lo, cr_hi
  int cr_lo, cr_hi;
};
static bool wave_threshold_params(int S, int vec, int M, int* gsz_out, int* kq_out, int* bits_out, ThresholdCheck* chk,
                                  int cand_cap = kCandCap) {
#ifndef PLA_CAND_MULT
#define PLA_CAND_MULT 2.2
#endif
  const double target = PLA_CAND_MULT * (M + 1);
  if (target > 0.75 * cand_cap || target >= 0.5 * S) return false;  // (the threshold is verified by an exact count before the sweep)
  const double F = 1.0 - target / S;
  const int S0 = S < kWave * kWaveSlots ? S : kWave * kWaveSlots;
  const int qfull = S0 / vec / kWave;  // vectors that are real draws in every lane
  const int nq = kWaveSlots / vec;
  const int bits = sample_bits_for(qfull, nq == 32 ? 5 : 4);
  int best_g = 0, best_k = 0;
  for (int g = 4; g <= 32; g <<= 1) {
    // the first g slots in visiting order are the vectors bitrev_order(j, bits), j < g / vec; those past the row are pads
    // (copies of the lane's first vector), so the group maximum is effectively over g_eff slots
    int g_eff = 0;
    for (int j = 0; j < (g + vec - 1) / vec; ++j)
      if (bitrev_order(j, bits) < qfull) g_eff += (g < vec ? g : vec);
    if (g_eff < 2) continue;
    const int k = (int)std::lround(kWave * std::pow(F, g_eff));
    if (k >= 6 && k <= 40 && (best_g == 0 || k > best_k)) { best_g = g; best_k = k; }
  }
  if (!best_g) return false;
  *gsz_out = best_g;
  *kq_out = best_k;
  *bits_out = bits;
  // acceptance band of the threshold check, in draws of the register block (S0 of the row's S; pads never count): the whole
  // row must end with M + 1 .. cand_cap draws above the threshold; 25 % / 15 % of margin for what the first chunk of a long
  // row cannot know
  static const int off = debug_flag("PLA_NO_THRESHOLD_CHECK");
  chk->cr_lo = (int)std::ceil(1.25 * (M + 1) / S * S0);
  chk->cr_hi = off ? 0 : (int)std::floor(0.85 * cand_cap / S * S0);
  return true;
}

static int debug_flag(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}

// Workgroups of a wave-per-row kernel with `waves` waves per workgroup: 512 workgroups are resident (2 per CU), so the grid is
// 512 x 2^k -- whole rounds -- with k as large as leaves every wave >= 24 rows (a wave's first row is loaded without overlap:
// short-lived waves pay that start-up again and again), at most 8 rounds (many rounds even out clock and memory-channel luck)
static int64_t wave_grid(int64_t n_obs, int waves) {
  const int64_t need = (n_obs + waves - 1) / waves;
  if (need <= 512) return need < 1 ? 1 : need;
  int64_t grid = 512;
  while (grid < 4096 && n_obs / (2 * grid * waves) >= 24) grid *= 2;
  return grid;
}

// second kernel of a split LOO pass: fit / smoothing / outputs for the tails the selection kernel handed over (pla_fit.h)
static hipError_t launch_fit(const RowsParams& p, const FastParams& f, int mestM, hipStream_t stream, const unsigned* fitted = nullptr,
                             const unsigned* gave_up = nullptr) {
  FitParams q{p.ws_y, p.ws_s, p.ws_stride, p.n_obs, p.n_draws, p.tail_count, mestM, f.log_S, p.scale_value, p.l1_table,
              p.l1_table + p.tail_count, p.diag, p.loo_i, p.lppd_i, p.slow_list, p.counters};
  q.slow_base = f.slow_base;
  q.ws_sstride = f.ws_sstride;
  q.fitted = const_cast<unsigned*>(fitted);  // (behind a streamed pass: only the chunks that pass left)
  q.gave_up = const_cast<unsigned*>(gave_up);
  static const int skip_fit = debug_flag("PLA_SKIP_FIT");  // timing experiments only: the outputs are then garbage
  if (skip_fit) return hipSuccess;
  const int nq = p.ws_stride / 64;
  const int waves = nq <= 4 ? kFitWaves : 2;
  int64_t g3 = ((p.n_obs + 3) / 4 + waves - 1) / waves;  // four observations per wave
  if (g3 > 256 * 8) g3 = 256 * 8;
  const dim3 fg((unsigned)g3), fb(kWave * waves);
  switch (nq) {
    case 1: hipLaunchKernelGGL(fit_rows_kernel<1>, fg, fb, 0, stream, q); break;
    case 2: hipLaunchKernelGGL(fit_rows_kernel<2>, fg, fb, 0, stream, q); break;
    case 3: hipLaunchKernelGGL(fit_rows_kernel<3>, fg, fb, 0, stream, q); break;
    case 4: hipLaunchKernelGGL(fit_rows_kernel<4>, fg, fb, 0, stream, q); break;
    case 5: hipLaunchKernelGGL((fit_rows_kernel<5, 4, 2>), fg, fb, 0, stream, q); break;
    case 6: hipLaunchKernelGGL((fit_rows_kernel<6, 4, 2>), fg, fb, 0, stream, q); break;
    default: hipLaunchKernelGGL((fit_rows_kernel<7, 4, 2>), fg, fb, 0, stream, q); break;
  }
  return hipGetLastError();
}
// Streamed pass, the fit kernel that runs beside the wave kernel: ONE four-wave workgroup per CU is what fits there (128
// registers per lane, 39.5 KB of LDS next to two workgroups of the wave kernel), and the workgroups stay for the whole launch
static hipError_t launch_fit_stream(const RowsParams& p, const FastParams& f, int mestM, unsigned* sync, hipStream_t stream,
                                    bool helper = false) {
  FitParams q{p.ws_y, p.ws_s, p.ws_stride, p.n_obs, p.n_draws, p.tail_count, mestM, f.log_S, p.scale_value, p.l1_table,
              p.l1_table + p.tail_count, p.diag, p.loo_i, p.lppd_i, p.slow_list, p.counters};
  q.slow_base = f.slow_base;
  q.ws_sstride = f.ws_sstride;
  const int64_t nchunks = (p.n_obs + kQueueChunk - 1) / kQueueChunk;
  q.take = sync + 16;
  q.gave_up = sync + 32;
  q.done = sync + 48;
  q.fitted = sync + 48 + nchunks;
  // looks are 1-2 us apart (s_sleep 32 + one load from memory): a few seconds of them, then the chunk is left to the plain fit
  // kernel that follows -- nothing then depends on the two kernels having been run side by side
  const int patience = debug_flag("PLA_STREAM_PATIENCE");  // (read per call: a test lets the fit kernel give up at once)
  q.patience = patience > 0 ? (unsigned)patience : 2000000u;
  static const int fg_forced = debug_flag("PLA_FIT_GRID");
  int64_t g = nchunks < 256 ? nchunks : 256;
  if (fg_forced > 0 && fg_forced < nchunks) g = fg_forced;
  if (helper) {
    // a second launch of the same kernel BEHIND the wave kernel on its stream: the fit kernel beside the wave kernel lives on
    // what that kernel leaves it and ends a few per cent of the chunks behind; once the wave kernel is gone these workgroups
    // (three more per CU) take chunks from the same counter and the tail is over in a few trips
    static const int hg = debug_flag("PLA_FIT_HELPERS");
    g = hg > 0 ? hg : (hg < 0 ? 0 : 768);
    if (g > nchunks) g = nchunks;
    if (g < 1) return hipSuccess;
  }
  const dim3 sg((unsigned)g), sb(kWave * 4);
  switch (p.ws_stride / 64) {  // (the coefficient scratch is dynamic LDS: pla_fit.h, DYN)
    case 1: hipLaunchKernelGGL(fit_rows_stream_kernel<1>, sg, sb, (fit_coef_bytes<1, 4>()), stream, q); break;
    case 2: hipLaunchKernelGGL(fit_rows_stream_kernel<2>, sg, sb, (fit_coef_bytes<2, 4>()), stream, q); break;
    case 3: hipLaunchKernelGGL(fit_rows_stream_kernel<3>, sg, sb, (fit_coef_bytes<3, 4>()), stream, q); break;
    default: hipLaunchKernelGGL(fit_rows_stream_kernel<4>, sg, sb, (fit_coef_bytes<4, 4>()), stream, q); break;
  }
  return hipGetLastError();
}
// zeroes two regions in one launch (the counters and the flags of a streamed pass: one command on the stream instead of two)
__global__ __launch_bounds__(256) void zero2_kernel(unsigned* a, size_t na, unsigned* b, size_t nb) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = i0; i < na; i += step) a[i] = 0u;
  for (size_t i = i0; i < nb; i += step) b[i] = 0u;
}
size_t stream_sync_bytes(int64_t n_obs) {
  const int64_t nchunks = (n_obs + kQueueChunk - 1) / kQueueChunk;
  return (size_t)(48 + 2 * nchunks + 16) * sizeof(unsigned);
}
// shapes the split pass covers: hand-over buffers present, tail within the stride, grid within the fit kernel's lanes
static bool split_ok(const RowsParams& p, int mestM) {
  if (!p.ws_y || !p.ws_s || p.ws_stride % 64 != 0 || p.tail_count > p.ws_stride) return false;
  return p.ws_stride <= 256 ? mestM <= kFitGrid : (p.ws_stride <= 448 && mestM <= kFitGridBig);
}

template <typename T, int VEC, bool LW>
static hipError_t launch_wave(const RowsParams& p, int gsz, int kq, int bits, const ThresholdCheck& chk, hipStream_t stream,
                              hipEvent_t after_first, bool* recorded, const PipeStreams* pipe, int* plan_stream) {
  hipError_t e = hipSuccess;
  static const int dbg = debug_flag("PLA_DEBUG_SKIP");
  static const int fused = debug_flag("PLA_FUSED");  // 1: single fused kernel (the pre-split pass), for A/B runs
  int root_ = (int)std::sqrt((double)p.tail_count);
  while (root_ * root_ > p.tail_count) --root_;
  while ((root_ + 1) * (root_ + 1) <= p.tail_count) ++root_;
  const int mestM = 30 + root_;
  FastParams f{gsz, kq, p.slow_list, p.counters, dbg, p.l1_table, std::log((double)p.n_draws), p.l1_table + p.tail_count, mestM};
  f.sample_bits = bits;
  f.cr_lo = chk.cr_lo; f.cr_hi = chk.cr_hi;
  // 4 independent waves per workgroup (they share the read-only tables); 8 x 2048 waves keep all
  // 256 CUs (8 waves each) busy with a short tail
  const int64_t grid = wave_grid(p.n_obs, kWavesPerBlock);
  const bool split = !LW && !fused && !(dbg & 31) && p.ws_stride <= 256 && split_ok(p, mestM);  // (ablation bits >= 32 live inside the split pass)
  const bool streamed = pipe && split && pipe->sync && p.ws_sstride == 16 && p.n_obs < ((int64_t)1 << 31);
  if (plan_stream) {
    *plan_stream = streamed ? 1 : 0;
    return hipSuccess;
  }
  if (!streamed) {
    e = hipMemsetAsync(p.counters, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
  }
  if constexpr (!LW) {
    if (split) {
      // split pass: wave kernel up to the exact selection, then sixteen lanes per observation for the GPD fit,
      // the smoothing sums and the outputs (pla_fit.h)
      f.ws_y = p.ws_y;
      f.ws_s = p.ws_s;
      f.ws_stride = p.ws_stride;
      f.ws_sstride = p.ws_sstride;
      if (streamed) {
        // streamed: the fit kernel runs beside the wave kernel and takes the chunks as they are finished
        unsigned* const sync = pipe->sync;
        const int64_t nchunks = (p.n_obs + kQueueChunk - 1) / kQueueChunk;
        {
          const size_t nsync = stream_sync_bytes(p.n_obs) / sizeof(unsigned);
          const unsigned zg = (unsigned)((nsync + 1023) / 1024 < 256 ? (nsync + 1023) / 1024 : 256);
          hipLaunchKernelGGL(zero2_kernel, dim3(zg ? zg : 1), dim3(256), 0, stream, reinterpret_cast<unsigned*>(p.counters),
                             (size_t)(pipe->zero_all_counters ? 32 : 2), sync, nsync);
          e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(pipe->fork, stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(pipe->first, pipe->fork, 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(pipe->second, pipe->fork, 0);
        if (e != hipSuccess) return e;
        f.queue = sync;
        f.done = sync + 48;
        // the wave kernel's waves issue first (s_setprio 3: A/B on C3 6.55 ms against 7.15 at equal priority); the fit kernel
        // fills what they leave and still ends with them (one workgroup per CU is ~70 % busy)
        static const char* wprio = getenv("PLA_WAVE_PRIO");
        f.prio = wprio ? atoi(wprio) : 3;
        const int64_t need = (p.n_obs + kWavesPerBlock * kQueueChunk - 1) / (kWavesPerBlock * kQueueChunk);
        const int64_t g1 = need < 512 ? need : 512;  // two resident workgroups per CU take everything there is from the queue
        note_kernels("wave_loo_kernel<%s, SPLIT, SYNC> (statistics, sweep, tail selection; dynamic row queue) with "
                     "fit_rows_stream_kernel<%d> beside it on a second stream (GPD fit, smoothing, outputs; chunks taken behind "
                     "per-chunk flags) + fit_rows_kernel (leftovers) + slow_rows_kernel (declined rows)", tname<T>(), p.ws_stride / 64);
        if (pipe->before_first) (void)hipEventRecord(pipe->before_first, pipe->first);
        hipLaunchKernelGGL((wave_loo_kernel<T, VEC, false, CapsSmall, true, true>), dim3((unsigned)g1), dim3(kWave * kWavesPerBlock), 0,
                           pipe->first, p, f);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (pipe->after_first) (void)hipEventRecord(pipe->after_first, pipe->first);
        e = launch_fit_stream(p, f, mestM, sync, pipe->second);
        if (e == hipSuccess) e = launch_fit_stream(p, f, mestM, sync, pipe->first, true);
        if (e == hipSuccess) e = hipEventRecord(pipe->join_first, pipe->first);
        if (e == hipSuccess) e = hipEventRecord(pipe->join_second, pipe->second);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, pipe->join_first, 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, pipe->join_second, 0);
        if (e != hipSuccess) return e;
        // whatever the streamed fit left (nothing, unless it gave up waiting for the wave kernel)
        e = launch_fit(p, f, mestM, stream, sync + 48 + nchunks, sync + 32);
        if (e != hipSuccess) return e;
      } else {
        const int64_t g1 = grid;
        note_kernels("wave_loo_kernel<%s, SPLIT> (statistics, sweep, tail selection) + fit_rows_kernel<%d> (GPD fit, smoothing, "
                     "outputs) + slow_rows_kernel (declined rows), back to back", tname<T>(), p.ws_stride / 64);
        hipLaunchKernelGGL((wave_loo_kernel<T, VEC, false, CapsSmall, true>), dim3((unsigned)g1), dim3(kWave * kWavesPerBlock), 0,
                           stream, p, f);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (after_first && hipEventRecord(after_first, stream) == hipSuccess && recorded) *recorded = true;
        e = launch_fit(p, f, mestM, stream);
        if (e != hipSuccess) return e;
      }
    }
  }
  if (!split) {
    note_kernels("wave_loo_kernel<%s%s> (fused: selection, fit and outputs in the wave) + slow_rows_kernel", tname<T>(), LW ? ", weights" : "");
    hipLaunchKernelGGL((wave_loo_kernel<T, VEC, LW, CapsSmall>), dim3((unsigned)grid), dim3(kWave * kWavesPerBlock), 0, stream, p, f);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // general kernel over whatever the fast path declined (usually nothing)
  constexpr int BLOCK = 256;
  int64_t g2 = p.n_obs < 1024 ? p.n_obs : 1024;
  hipLaunchKernelGGL((slow_rows_kernel<T, BLOCK, LW>), dim3((unsigned)g2), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap),
                     stream, p);
  return hipGetLastError();
}

// long rows (chunks of 4096 draws) and / or tail counts up to 512: pla_chunked.h
template <typename T, int VEC, class CAP, bool LW = false>
static hipError_t launch_chunked(const RowsParams& p, int gsz, int kq, int bits, const ThresholdCheck& chk, hipStream_t stream,
                                 hipEvent_t after_first, bool* recorded) {
  hipError_t e = hipSuccess;
  static const int fused = debug_flag("PLA_FUSED");
  int root_ = (int)std::sqrt((double)p.tail_count);
  while (root_ * root_ > p.tail_count) --root_;
  while ((root_ + 1) * (root_ + 1) <= p.tail_count) ++root_;
  const int mestM = 30 + root_;
  FastParams f{gsz, kq, p.slow_list, p.counters, 0, p.l1_table, std::log((double)p.n_draws), p.l1_table + p.tail_count, mestM};
  f.sample_bits = bits;
  f.cr_lo = chk.cr_lo; f.cr_hi = chk.cr_hi;
  {
    // rows whose first-chunk threshold misses come round again (pla_chunked.h, ChunkRetry): what a later attempt aims at is the
    // middle of what the list accepts.  PLA_NO_RETRY=1: such rows go to the general kernel as in round 2 (A/B runs).
    static const int no_retry = debug_flag("PLA_NO_RETRY");
    const int nch = (p.n_draws + kChunkDraws - 1) / kChunkDraws;
    f.retry_target = (no_retry || nch < 1) ? 0 : (int)(0.5 * ((p.tail_count + 1) + (double)CAP::kCand));
  }
  constexpr int W = CAP::kWaves;  // waves per workgroup; two workgroups per CU (LDS)
  int64_t grid = (p.n_obs + W - 1) / W;
  if (grid > 2048 * 8 / W) grid = 2048 * 8 / W;
  bool split = false;
  if constexpr (CAP::kMaxTail <= 448 && !LW) split = !fused && split_ok(p, mestM) && p.ws_stride <= 64 * ((CAP::kMaxTail + 63) / 64);
  e = hipMemsetAsync(p.counters, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  if (split) {
    if constexpr (CAP::kMaxTail <= 448 && !LW) {
      f.ws_y = p.ws_y;
      f.ws_s = p.ws_s;
      f.ws_stride = p.ws_stride;
      note_kernels("wave_loo_chunked_kernel<%s, SPLIT> (rows in chunks of 4096 draws: statistics, sweep, tail selection) + "
                   "fit_rows_kernel<%d> (GPD fit, smoothing, outputs) + slow_rows_kernel (declined rows), back to back", tname<T>(),
                   p.ws_stride / 64);
      hipLaunchKernelGGL((wave_loo_chunked_kernel<T, VEC, CAP, true>), dim3((unsigned)grid), dim3(kWave * W), 0, stream, p, f);
      e = hipGetLastError();
      if (e != hipSuccess) return e;
      if (after_first && hipEventRecord(after_first, stream) == hipSuccess && recorded) *recorded = true;
      e = launch_fit(p, f, mestM, stream);
      if (e != hipSuccess) return e;
    }
  } else {
    note_kernels("wave_loo_chunked_kernel<%s%s> (fused) + slow_rows_kernel", tname<T>(), LW ? ", weights" : "");
    hipLaunchKernelGGL((wave_loo_chunked_kernel<T, VEC, CAP, false, LW>), dim3((unsigned)grid), dim3(kWave * W), 0, stream, p, f);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // the rows this path declined (C5: the ~0.1 % behind the cancellation guard).  They are few, so what they cost is the
  // LATENCY of one workgroup walking one long row thirteen times: 1024 threads per row instead of 256 for rows of 8192+ draws
  int64_t g2 = p.n_obs < 1024 ? p.n_obs : 1024;
  if (p.n_draws >= 8192) {
    constexpr int BLOCK = 1024;
    hipLaunchKernelGGL((slow_rows_kernel<T, BLOCK, LW>), dim3((unsigned)g2), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap), stream, p);
  } else {
    constexpr int BLOCK = 256;
    hipLaunchKernelGGL((slow_rows_kernel<T, BLOCK, LW>), dim3((unsigned)g2), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap), stream, p);
  }
  return hipGetLastError();
}

// SIS / TIS in LOO mode: streaming kernel (pla_is.h) + the general kernel for the rows it declines
template <typename T, int VEC, bool LW = false>
static hipError_t launch_is(const RowsParams& p, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(p.counters, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  FastParams f{0, 0, p.slow_list, p.counters, 0, nullptr, std::log((double)p.n_draws), nullptr, 0};
  int64_t grid = (p.n_obs + kWavesPerBlock - 1) / kWavesPerBlock;
  if (grid > 2048 * 8 / kWavesPerBlock) grid = 2048 * 8 / kWavesPerBlock;
  note_kernels("is_wave_kernel<%s, %s%s> + slow_rows_kernel", tname<T>(), p.method == PLA_TIS ? "TIS" : "SIS", LW ? ", weights" : "");
  if (p.method == PLA_TIS)
    hipLaunchKernelGGL((is_wave_kernel<T, VEC, true, LW>), dim3((unsigned)grid), dim3(kWave * kWavesPerBlock), 0, stream, p, f);
  else
    hipLaunchKernelGGL((is_wave_kernel<T, VEC, false, LW>), dim3((unsigned)grid), dim3(kWave * kWavesPerBlock), 0, stream, p, f);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  constexpr int BLOCK = 256;
  int64_t g2 = p.n_obs < 1024 ? p.n_obs : 1024;
  hipLaunchKernelGGL((slow_rows_kernel<T, BLOCK, LW>), dim3((unsigned)g2), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap),
                     stream, p);
  return hipGetLastError();
}

template <typename T, bool LW>
static hipError_t launch_typed(const RowsParams& p, hipStream_t stream, hipEvent_t after_first, bool* recorded,
                               const PipeStreams* pipe = nullptr, int* plan = nullptr) {
  if (plan) *plan = 0;
  constexpr int BLOCK = 256;
  const bool unit = (p.stride_draw == 1);
  {
    static const int path = debug_flag("PLA_FORCE_PATH");  // 0 auto, 1 general kernel only (tests)
    constexpr int WVEC = 16 / sizeof(T);
    bool waligned = ((uintptr_t)p.in % 16 == 0) && (p.stride_obs % WVEC == 0) && (p.n_draws % WVEC == 0);
    if (LW) waligned = waligned && ((uintptr_t)p.lw_out % 16 == 0);  // rows of lw_out are n_draws apart
    if (path != 1 && unit && waligned && p.method == PLA_PSIS && p.slow_list && p.counters && p.l1_table &&
        p.n_draws <= kWave * kWaveSlots && p.n_draws >= 256 && p.tail_count <= kWaveMaxTail &&
        smem_bytes(BLOCK, p.tail_cap) <= 64 * 1024 && p.n_obs <= 0xffffffffll) {
      int gsz = 0, kq = 0, bits = 0;
      ThresholdCheck chk{};
      if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk)) {
        return launch_wave<T, WVEC, LW>(p, gsz, kq, bits, chk, stream, after_first, recorded, pipe, plan);
      }
    }
    if (path != 1 && unit && waligned && (p.method == PLA_SIS || p.method == PLA_TIS) && p.slow_list && p.counters &&
        p.n_draws <= kWave * kWaveSlots && p.n_draws >= kWave * WVEC && p.n_obs <= 0xffffffffll)
      return plan ? hipSuccess : launch_is<T, WVEC, LW>(p, stream);
    {
      // rows beyond one register chunk or tails beyond the small kernel's LDS: the chunked kernel
      // (weights mode: the candidates carry 16-bit draw indices, so rows up to 65 536 draws; two LDS capacities)
      const int last_chunk = p.n_draws - ((p.n_draws - 1) / kChunkDraws) * kChunkDraws;
      if (path != 1 && unit && waligned && p.method == PLA_PSIS && p.slow_list && p.counters && p.l1_table &&
          p.n_draws >= 256 && p.n_draws <= (1 << 20) && last_chunk >= kWave * WVEC && p.tail_count <= CapsBig::kMaxTail &&
          (!LW || p.n_draws <= 65536) && smem_bytes(BLOCK, p.tail_cap) <= 64 * 1024 && p.n_obs <= 0xffffffffll) {
        int gsz = 0, kq = 0, bits = 0;
        ThresholdCheck chk{};
        if constexpr (LW) {
          // (f32 rows: six waves per CU; f64 rows need more than 256 registers per lane next to the row, so four)
          using CapLW = std::conditional_t<sizeof(T) == 4, CapsMidLW, CapsMid>;
          if (p.tail_count <= CapLW::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapLW::kCand))
            return plan ? hipSuccess : launch_chunked<T, WVEC, CapLW, true>(p, gsz, kq, bits, chk, stream, after_first, recorded);
          if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsBig::kCand))
            return plan ? hipSuccess : launch_chunked<T, WVEC, CapsBig, true>(p, gsz, kq, bits, chk, stream, after_first, recorded);
        } else {
        if (p.tail_count <= CapsMid4::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsMid4::kCand))
          return plan ? hipSuccess : launch_chunked<T, WVEC, CapsMid4>(p, gsz, kq, bits, chk, stream, after_first, recorded);
        if (p.tail_count <= CapsMid::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsMid::kCand))
          return plan ? hipSuccess : launch_chunked<T, WVEC, CapsMid>(p, gsz, kq, bits, chk, stream, after_first, recorded);
        if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsBig::kCand))
          return plan ? hipSuccess : launch_chunked<T, WVEC, CapsBig>(p, gsz, kq, bits, chk, stream, after_first, recorded);
        }
      }
    }
    if (plan) return hipSuccess;
    if (path == 1) return launch_one<T, BLOCK, 0, LW>(p, stream);
  }
  if (unit && p.n_draws <= BLOCK * 16 && p.n_draws > BLOCK * 4) return launch_one<T, BLOCK, 16, LW>(p, stream);
  return launch_one<T, BLOCK, 0, LW>(p, stream);
}

hipError_t launch_rows(const RowsParams& p, int dtype, bool lw_mode, hipStream_t stream, hipEvent_t after_first, bool* recorded,
                       const PipeStreams* pipe) {
  if (p.n_obs <= 0) return hipSuccess;
  if (dtype == PLA_F64)
    return lw_mode ? launch_typed<double, true>(p, stream, after_first, recorded, pipe) : launch_typed<double, false>(p, stream, after_first, recorded, pipe);
  return lw_mode ? launch_typed<float, true>(p, stream, after_first, recorded, pipe) : launch_typed<float, false>(p, stream, after_first, recorded, pipe);
}

bool rows_stream_planned(const RowsParams& p, int dtype) {
  if (p.n_obs <= 0) return false;
  int plan = 0;
  PipeStreams probe{};
  probe.sync = reinterpret_cast<unsigned*>(p.counters);  // (any non-null pointer: looked at, not dereferenced)
  if (dtype == PLA_F64) (void)launch_typed<double, false>(p, nullptr, nullptr, nullptr, &probe, &plan);
  else (void)launch_typed<float, false>(p, nullptr, nullptr, nullptr, &probe, &plan);
  return plan != 0;
}

template <typename T>
static hipError_t launch_waic_typed(const WaicParams& p, hipStream_t stream) {
  constexpr int WVEC = 16 / sizeof(T);
  const bool fast = p.stride_draw == 1 && ((uintptr_t)p.in % 16 == 0) && (p.stride_obs % WVEC == 0) &&
                    (p.n_draws % WVEC == 0) && p.n_draws <= kWave * kWaveSlots && p.n_draws >= kWave * WVEC;
  static const int path = debug_flag("PLA_FORCE_PATH");
  if (fast && path != 1) {
    int64_t grid = (p.n_obs + kWavesPerBlock - 1) / kWavesPerBlock;
    if (grid > 2048 * 8 / kWavesPerBlock) grid = 2048 * 8 / kWavesPerBlock;
    hipLaunchKernelGGL((waic_wave_kernel<T, WVEC>), dim3((unsigned)grid), dim3(kWave * kWavesPerBlock), 0, stream, p);
  } else {
    int64_t grid = p.n_obs < 8192 ? p.n_obs : 8192;
    hipLaunchKernelGGL((waic_rows_kernel<T, 256>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  }
  return hipGetLastError();
}

// 64 (observations) x TD (draws) tiles through LDS: 512-byte segments when reading (lanes along the observations), 8 TD bytes
// when writing (lanes along the draws); the +1 pitch keeps the column reads at two lanes per bank for f64 and conflict-free
// for f32
#ifndef PLA_TRANSPOSE_TD
#define PLA_TRANSPOSE_TD 16  // 16 draws per tile: 128-byte writes, 8 KB of LDS, more workgroups in flight (4.65 TB/s against 4.36 at 64)
#endif
template <typename T, int TD>
__global__ __launch_bounds__(256) void transpose_rows_kernel(const T* __restrict__ in, int64_t stride_draw, int64_t obs0,
                                                             int64_t n_rows, int n_draws, T* __restrict__ out) {
  constexpr int TO = 64;
  __shared__ T tile[TD][TO + 1];
  const int64_t o0 = (int64_t)blockIdx.x * TO;
  const int d0 = (int)blockIdx.y * TD;
  {
    const int tx = threadIdx.x & (TO - 1), ty = threadIdx.x >> 6;
    const int64_t oi = o0 + tx;
#pragma unroll 4
    for (int d = ty; d < TD; d += 4) {
      const int dd = d0 + d;
      if (dd < n_draws && oi < n_rows) tile[d][tx] = __builtin_nontemporal_load(in + (int64_t)dd * stride_draw + obs0 + oi);
    }
  }
  __syncthreads();
  {
    const int tx = threadIdx.x & (TD - 1), ty = threadIdx.x / TD;
    const int dw = d0 + tx;
#pragma unroll 4
    for (int o = ty; o < TO; o += 256 / TD) {
      const int64_t oo = o0 + o;
      if (oo < n_rows && dw < n_draws) out[oo * n_draws + dw] = tile[tx][o];
    }
  }
}

hipError_t launch_transpose_rows(const void* in, int dtype, int64_t stride_draw, int64_t obs0, int64_t n_rows, int n_draws,
                                 void* out, hipStream_t stream) {
  if (n_rows <= 0 || n_draws <= 0) return hipSuccess;
  constexpr int TD = PLA_TRANSPOSE_TD;
  const dim3 grid((unsigned)((n_rows + 63) / 64), (unsigned)((n_draws + TD - 1) / TD));
  if (dtype == PLA_F64)
    hipLaunchKernelGGL((transpose_rows_kernel<double, TD>), grid, dim3(256), 0, stream, (const double*)in, stride_draw, obs0, n_rows,
                       n_draws, (double*)out);
  else
    hipLaunchKernelGGL((transpose_rows_kernel<float, TD>), grid, dim3(256), 0, stream, (const float*)in, stride_draw, obs0, n_rows,
                       n_draws, (float*)out);
  return hipGetLastError();
}

__global__ void clamp_rows_kernel(const int64_t* in, int64_t n_rows, int64_t n_src, int64_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = in[i];
    out[i] = v < 0 ? 0 : (v >= n_src ? n_src - 1 : v);
  }
}

hipError_t launch_clamp_rows(const int64_t* in, int64_t n_rows, int64_t n_src, int64_t* out, hipStream_t stream) {
  if (n_rows <= 0) return hipSuccess;
  int64_t g = (n_rows + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(clamp_rows_kernel, dim3((unsigned)g), dim3(256), 0, stream, in, n_rows, n_src, out);
  return hipGetLastError();
}

hipError_t launch_waic(const void* in, const int64_t* row_index, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs, int64_t stride_draw,
                       double scale_value, double* lppd_i, double* var_i, double* waic_i,
                       unsigned long long* replaced, hipStream_t stream) {
  if (n_obs <= 0) return hipSuccess;
  WaicParams p{in, n_obs, n_draws, stride_obs, stride_draw, scale_value, lppd_i, var_i, waic_i, replaced, row_index};
  return dtype == PLA_F64 ? launch_waic_typed<double>(p, stream) : launch_waic_typed<float>(p, stream);
}

bool col_supported(int n_draws, int tail_count, int* kq) {
  // threshold: the kq-th smallest of 64 maxima over groups of 8 sampled draws has a fraction F of the row below it,
  // F^8 = kq / 64; aim at 2.2 (M + 1) draws above it, as in the wave kernel
  if (n_draws < kColSample || tail_count > CapsSmall::kMaxTail) return false;
  const double F = 1.0 - 2.2 * (tail_count + 1) / n_draws;
  if (!(F > 0.5)) return false;
  const int k = (int)std::lround(64.0 * std::pow(F, 8));
  if (k < 4 || k > 56) return false;
  *kq = k;
  return true;
}
size_t col_workspace_bytes(int64_t n_obs) { return (size_t)((n_obs + 63) & ~63ll) * (kColCap + 8) * sizeof(double); }  // (lists in groups of 64)

hipError_t launch_col(const RowsParams& p, int dtype, int kq, void* col_ws, hipStream_t stream) {
  if (p.n_obs <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(p.counters, 0, sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  int root_ = (int)std::sqrt((double)p.tail_count);
  while (root_ * root_ > p.tail_count) --root_;
  while ((root_ + 1) * (root_ + 1) <= p.tail_count) ++root_;
  const int mestM = 30 + root_;
  ColParams c{p.in, p.n_obs, p.n_draws, p.stride_draw, kq, (double*)col_ws, (double*)col_ws + (size_t)((p.n_obs + 63) & ~63ll) * kColCap};
  const unsigned g1 = (unsigned)((p.n_obs + 255) / 256);
  if (dtype == PLA_F64) hipLaunchKernelGGL(col_sweep_kernel<double>, dim3(g1), dim3(256), 0, stream, c);
  else hipLaunchKernelGGL(col_sweep_kernel<float>, dim3(g1), dim3(256), 0, stream, c);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  FastParams f{0, 0, p.slow_list, p.counters, 0, p.l1_table, std::log((double)p.n_draws), p.l1_table + p.tail_count, mestM};
  f.ws_y = p.ws_y;
  f.ws_s = p.ws_s;
  f.ws_stride = p.ws_stride;
  int64_t g2 = (p.n_obs + 3) / 4;
  if (g2 > 256 * 16) g2 = 256 * 16;
  hipLaunchKernelGGL(col_select_kernel<CapsSmall>, dim3((unsigned)g2), dim3(kWave * 4), 0, stream, c, f, p.tail_count);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = launch_fit(p, f, mestM, stream);
  if (e != hipSuccess) return e;
  // rows the column path declined: the general kernel walks them with the matrix's strides
  constexpr int BLOCK = 256;
  int64_t g3 = p.n_obs < 1024 ? p.n_obs : 1024;
  if (dtype == PLA_F64)
    hipLaunchKernelGGL((slow_rows_kernel<double, BLOCK, false>), dim3((unsigned)g3), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap), stream, p);
  else
    hipLaunchKernelGGL((slow_rows_kernel<float, BLOCK, false>), dim3((unsigned)g3), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap), stream, p);
  return hipGetLastError();
}

// ---- observations-fastest LOO, a workgroup per 16 observations (pla_tile.h) -------------------------------------------------
// The threshold is the sample's ks-th largest value: ks / 512 of the row is expected at or above it, T = ks S / 512 draws, with a
// standard deviation of sqrt(S^2 p (1 - p) / 512 + S p (1 - p)), p = T / S (the sample's quantile, then the row's count given the
// quantile).  T sits halfway between the M + 1 the selection needs and the list's capacity; shapes where that leaves less than
// 3.0 standard deviations on either side stay with the lane-per-observation kernels (pla_col.h).
bool tile_supported(int dtype, int n_draws, int tail_count, int64_t ld, bool streamed, int* ks) {
  static const int off = debug_flag("PLA_NO_TILE");
  if (off || dtype != PLA_F64) return false;
  if (n_draws < kTileSample || tail_count > CapsSmall::kMaxTail || tail_count < 1) return false;
  if ((double)ld * 8.0 * 4.0 >= 4294967296.0) return false;  // the lane's draw inside a step rides in a 32-bit offset
  const int cap = streamed ? kTileCapStream : kTileCap;
  const double need = streamed ? 2.9 : 3.0;  // (the streamed pass has the shorter lists: it pays for itself down to here)
  const double target = 0.5 * (tail_count + 1 + cap);
  int k = (int)std::lround(target * kTileSample / n_draws);
  if (k < 2) return false;
  if (k > kTileSample - 1) k = kTileSample - 1;   // (short rows: nearly every draw is a candidate, and fits)
  const double p = (double)k / kTileSample, T = p * n_draws;
  const double sd = std::sqrt((double)n_draws * n_draws * p * (1 - p) / kTileSample + n_draws * p * (1 - p));
  if (n_draws > cap && (T - (tail_count + 1) < need * sd || cap - T < need * sd)) return false;
  if (n_draws <= cap && T - (tail_count + 1) < need * sd) return false;
  *ks = k;
  return true;
}

// (the kernel's LDS is beyond the 64 KB a launch may ask for by default; the attribute belongs to the device's copy of the
// function, so it is set once per device of the process -- callers hold the engine's mutex, engines of different devices may
// race for their own slot only)
template <bool SYNC>
static bool tile_lds_attr() {
  using SMT = TileSmem<double, SYNC ? kTileCapStream : kTileCap>;
  static std::atomic<int> state[64];  // per device: 0 not tried, 1 set, 2 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  int st = state[dev].load(std::memory_order_acquire);
  if (st == 0) {
    st = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_loo_kernel<double, SYNC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)sizeof(SMT)) == hipSuccess ? 1 : 2;
    state[dev].store(st, std::memory_order_release);
  }
  return st == 1;
}

hipError_t launch_tile(const RowsParams& p, int dtype, int ks, hipStream_t stream, const PipeStreams* pipe) {
  if (p.n_obs <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  int root_ = (int)std::sqrt((double)p.tail_count);
  while (root_ * root_ > p.tail_count) --root_;
  while ((root_ + 1) * (root_ + 1) <= p.tail_count) ++root_;
  const int mestM = 30 + root_;
  TileParams c{p.in, p.n_obs, p.n_draws, p.stride_draw, ks};
  FastParams f{0, 0, p.slow_list, p.counters, 0, p.l1_table, std::log((double)p.n_draws), p.l1_table + p.tail_count, mestM};
  f.ws_y = p.ws_y;
  f.ws_s = p.ws_s;
  f.ws_stride = p.ws_stride;
  f.ws_sstride = p.ws_sstride;
  const int64_t ngroups = (p.n_obs + 15) / 16;
  static const int grid_env = debug_flag("PLA_TILE_GRID");
  const int64_t cap = grid_env > 0 ? grid_env : 256;  // one workgroup per CU
  const unsigned g1 = (unsigned)(ngroups < cap ? ngroups : cap);
  const bool streamed = pipe && pipe->sync && p.ws_sstride == 16 && p.ws_stride <= 256 && p.n_obs < ((int64_t)1 << 31) && split_ok(p, mestM);
  if (streamed) {
    // streamed: the fit kernel runs beside the tile kernel and takes each group of 16 observations as its flag goes up
    // (launch_wave, streamed branch: the same flags, streams and leftovers)
    if (!tile_lds_attr<true>()) return hipErrorInvalidValue;
    unsigned* const sync = pipe->sync;
    const int64_t nchunks = (p.n_obs + kQueueChunk - 1) / kQueueChunk;
    {
      const size_t nsync = stream_sync_bytes(p.n_obs) / sizeof(unsigned);
      const unsigned zg = (unsigned)((nsync + 1023) / 1024 < 256 ? (nsync + 1023) / 1024 : 256);
      hipLaunchKernelGGL(zero2_kernel, dim3(zg ? zg : 1), dim3(256), 0, stream, reinterpret_cast<unsigned*>(p.counters),
                         (size_t)(pipe->zero_all_counters ? 32 : 2), sync, nsync);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipEventRecord(pipe->fork, stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(pipe->first, pipe->fork, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(pipe->second, pipe->fork, 0);
    if (e != hipSuccess) return e;
    f.queue = sync;  // (groups beyond a workgroup's first: sync[0], zero at launch)
    f.done = sync + 48;
    static const char* wprio = getenv("PLA_WAVE_PRIO");
    f.prio = wprio ? atoi(wprio) : 3;
    if (pipe->before_first) (void)hipEventRecord(pipe->before_first, pipe->first);
    hipLaunchKernelGGL((tile_loo_kernel<double, true>), dim3(g1), dim3(kTileThreads), sizeof(TileSmem<double, kTileCapStream>), pipe->first,
                       c, f, p.tail_count);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (pipe->after_first) (void)hipEventRecord(pipe->after_first, pipe->first);
    e = launch_fit_stream(p, f, mestM, sync, pipe->second);
    if (e == hipSuccess) e = launch_fit_stream(p, f, mestM, sync, pipe->first, true);
    if (e == hipSuccess) e = hipEventRecord(pipe->join_first, pipe->first);
    if (e == hipSuccess) e = hipEventRecord(pipe->join_second, pipe->second);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, pipe->join_first, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(stream, pipe->join_second, 0);
    if (e != hipSuccess) return e;
    // whatever the streamed fit left (nothing, unless it gave up waiting for the tile kernel)
    e = launch_fit(p, f, mestM, stream, sync + 48 + nchunks, sync + 32);
    if (e != hipSuccess) return e;
  } else {
    e = hipMemsetAsync(p.counters, 0, sizeof(unsigned long long), stream);
    if (e == hipSuccess) e = hipMemsetAsync(p.counters + 4, 0, sizeof(unsigned long long), stream);  // [4]: the kernel's group counter
    if (e != hipSuccess) return e;
    f.queue = reinterpret_cast<unsigned*>(p.counters + 4);
    if (!tile_lds_attr<false>()) return hipErrorInvalidValue;
    hipLaunchKernelGGL((tile_loo_kernel<double, false>), dim3(g1), dim3(kTileThreads), sizeof(TileSmem<double, kTileCap>), stream, c, f,
                       p.tail_count);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_fit(p, f, mestM, stream);
    if (e != hipSuccess) return e;
  }
  // rows the tile kernel declined: the general kernel walks them with the matrix's strides
  constexpr int BLOCK = 256;
  int64_t g3 = p.n_obs < 1024 ? p.n_obs : 1024;
  hipLaunchKernelGGL((slow_rows_kernel<double, BLOCK, false>), dim3((unsigned)g3), dim3(BLOCK), smem_bytes(BLOCK, p.tail_cap), stream, p);
  return hipGetLastError();
}

hipError_t launch_e_loo(const void* x, const void* lw, const void* lr, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs,
                        int64_t stride_draw, int tail_len, double* mean, double* var, double* k_mean, double* k_var, double* k_none,
                        unsigned* slow_list, unsigned long long* slow_count, hipStream_t stream) {
  if (n_obs <= 0) return hipSuccess;
  ELooParams p{x, lw, lr ? lr : lw, n_obs, n_draws, stride_obs, stride_draw, tail_len, mean, var, k_mean, k_var, k_none, slow_list, slow_count};
  const int64_t grid = n_obs < 16384 ? n_obs : 16384;
  static const int path = debug_flag("PLA_FORCE_PATH");  // 1: general kernel only (tests)
  const int vec = dtype == PLA_F64 ? 2 : 4;
  const uintptr_t align = (uintptr_t)x | (uintptr_t)lw | (uintptr_t)(lr ? lr : lw);
  // wave per observation, one pass (pla_eloo.h): contiguous draws in 16-byte vectors, rows the list can name
  if (path != 1 && slow_list && slow_count && stride_draw == 1 && align % 16 == 0 && stride_obs % vec == 0 && n_draws % vec == 0 &&
      n_draws >= kWave * vec && n_draws <= (1 << 20) && n_obs <= 0xffffffffll) {
    hipError_t e = hipMemsetAsync(slow_count, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    int64_t wg = (n_obs + 3) / 4;
    if (wg > 8192) wg = 8192;
    const bool own = lr && lr != lw;
    if (dtype == PLA_F64) {
      if (own) hipLaunchKernelGGL((e_loo_wave_kernel<double, true>), dim3((unsigned)wg), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((e_loo_wave_kernel<double, false>), dim3((unsigned)wg), dim3(256), 0, stream, p);
    } else {
      if (own) hipLaunchKernelGGL((e_loo_wave_kernel<float, true>), dim3((unsigned)wg), dim3(256), 0, stream, p);
      else hipLaunchKernelGGL((e_loo_wave_kernel<float, false>), dim3((unsigned)wg), dim3(256), 0, stream, p);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int64_t g2 = n_obs < 2048 ? n_obs : 2048;  // the rows it declined (NaN / inf): usually none
    if (dtype == PLA_F64) hipLaunchKernelGGL((e_loo_rows_kernel<double, 256, true>), dim3((unsigned)g2), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((e_loo_rows_kernel<float, 256, true>), dim3((unsigned)g2), dim3(256), 0, stream, p);
    return hipGetLastError();
  }
  if (dtype == PLA_F64) hipLaunchKernelGGL((e_loo_rows_kernel<double, 256>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((e_loo_rows_kernel<float, 256>), dim3((unsigned)grid), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_e_loo_quantiles(const void* x, const void* lw, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs,
                                  int64_t stride_draw, const double* probs, int n_probs, double* out, unsigned* slow_list,
                                  unsigned long long* slow_count, hipStream_t stream) {
  if (n_obs <= 0 || n_probs <= 0) return hipSuccess;
  EQuantParams p{x, lw, n_obs, n_draws, stride_obs, stride_draw, probs, n_probs, out};
  const int64_t grid = n_obs < 16384 ? n_obs : 16384;
  // 512 threads per observation, eight draws per thread in registers.  Rows of up to 4096 draws go through the FAST variant
  // (histogram path only: 128 registers, two workgroups per CU) and the few it lists through the general one behind it.
  const bool two = slow_list && slow_count && n_draws <= 4096 && n_obs <= 0xffffffffll;
  if (two) {
    hipError_t e = hipMemsetAsync(slow_count, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    p.slow_list = slow_list;
    p.slow_count = slow_count;
    if (dtype == PLA_F64) hipLaunchKernelGGL((e_loo_quantile_kernel<double, 512, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((e_loo_quantile_kernel<float, 512, true>), dim3((unsigned)grid), dim3(512), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int64_t g2 = two ? (n_obs < 2048 ? n_obs : 2048) : grid;
  if (dtype == PLA_F64) hipLaunchKernelGGL((e_loo_quantile_kernel<double, 512>), dim3((unsigned)g2), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((e_loo_quantile_kernel<float, 512>), dim3((unsigned)g2), dim3(512), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_waic_col(const void* in, int dtype, int64_t n_obs, int n_draws, int64_t ld, double scale_value, double* lppd_i,
                           double* var_i, double* waic_i, unsigned long long* replaced, hipStream_t stream) {
  if (n_obs <= 0) return hipSuccess;
  WaicParams p{in, n_obs, n_draws, 1, ld, scale_value, lppd_i, var_i, waic_i, replaced, nullptr};
  const unsigned grid = (unsigned)((n_obs + 255) / 256);
  if (dtype == PLA_F64) hipLaunchKernelGGL(waic_col_kernel<double>, dim3(grid), dim3(256), 0, stream, p, ld);
  else hipLaunchKernelGGL(waic_col_kernel<float>, dim3(grid), dim3(256), 0, stream, p, ld);
  return hipGetLastError();
}

int reduce_workspace_doubles() { return kRedChunks * kRedSlots; }

hipError_t launch_reduce(const ReduceParams& p, double* workspace, hipStream_t stream) {
  int64_t chunks = (p.n_obs + 1023) / 1024;  // >= 1024 observations per chunk
  if (chunks < 1) chunks = 1;
  if (chunks > kRedChunks) chunks = kRedChunks;
  hipLaunchKernelGGL(reduce_stage1, dim3((unsigned)chunks), dim3(kRedBlock), 0, stream, p, workspace);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(kWave), 0, stream, p, workspace, (int)chunks);
  return hipGetLastError();
}

__global__ __launch_bounds__(kWave) void aggregate_pack_kernel(const double* agg, int rank, int world, double* table) {
  for (int i = threadIdx.x; i < world * kRedSlots; i += kWave) table[i] = (i / kRedSlots == rank) ? agg[i % kRedSlots] : 0.0;
}
// (one lane: a rank count is a handful; the order of the merge is fixed, so every rank gets the same bits)
__global__ __launch_bounds__(kWave) void aggregate_merge_kernel(const double* table, int world, double* out) {
  if (threadIdx.x != 0) return;
  Moments a{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, pinf()};
  double n_slow = 0.0;
  for (int r = 0; r < world; ++r) {
    const double* o = table + (size_t)r * kRedSlots;
    Moments b{o[PLA_AGG_N], o[PLA_AGG_N] > 0.0 ? o[PLA_AGG_SUM_LOO] / o[PLA_AGG_N] : 0.0, o[PLA_AGG_M2_LOO], o[PLA_AGG_SUM_LOO],
              o[PLA_AGG_SUM_LPPD], o[PLA_AGG_N_HIGH], o[PLA_AGG_N_NONFINITE], o[PLA_AGG_MIN_DIAG]};
    merge_moments(a, b);
    n_slow += o[PLA_AGG_N_SLOW];
  }
  out[PLA_AGG_N] = a.n;
  out[PLA_AGG_SUM_LOO] = a.s_loo;
  out[PLA_AGG_M2_LOO] = a.m2;
  out[PLA_AGG_SUM_LPPD] = a.s_lppd;
  out[PLA_AGG_N_HIGH] = a.n_high;
  out[PLA_AGG_N_NONFINITE] = a.n_bad;
  out[PLA_AGG_MIN_DIAG] = a.dmin;
  out[PLA_AGG_N_SLOW] = n_slow;
}
hipError_t launch_aggregate_pack(const double* agg, int rank, int world, double* table, hipStream_t stream) {
  hipLaunchKernelGGL(aggregate_pack_kernel, dim3(1), dim3(kWave), 0, stream, agg, rank, world, table);
  return hipGetLastError();
}
hipError_t launch_aggregate_merge(const double* table, int world, double* out, hipStream_t stream) {
  hipLaunchKernelGGL(aggregate_merge_kernel, dim3(1), dim3(kWave), 0, stream, table, world, out);
  return hipGetLastError();
}

hipError_t launch_fill_chains(void* ll, int dtype, int64_t n_obs, int64_t n_draws, int chains, double rho, double off_sd, int64_t row0,
                              uint64_t seed, double k_lo, double k_hi, hipStream_t stream) {
  if (n_obs <= 0 || n_draws <= 0) return hipSuccess;
  const int64_t total = n_obs * chains;
  const unsigned grid = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
  if (dtype == PLA_F64)
    hipLaunchKernelGGL(fill_chains_kernel<double>, dim3(grid), dim3(256), 0, stream, (double*)ll, n_obs, n_draws, chains, rho, off_sd,
                       row0, seed, k_lo, k_hi);
  else
    hipLaunchKernelGGL(fill_chains_kernel<float>, dim3(grid), dim3(256), 0, stream, (float*)ll, n_obs, n_draws, chains, rho, off_sd,
                       row0, seed, k_lo, k_hi);
  return hipGetLastError();
}

hipError_t launch_fill_synthetic(void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t row0,
                                 uint64_t seed, double k_lo, double k_hi, double heavy_lo,
                                 double heavy_hi, hipStream_t stream) {
  if (n_obs <= 0 || n_draws <= 0) return hipSuccess;
  const unsigned grid = 256 * 16;
  if (dtype == PLA_F64)
    hipLaunchKernelGGL(fill_kernel<double>, dim3(grid), dim3(256), 0, stream, (double*)ll, n_obs, n_draws,
                       row0, seed, k_lo, k_hi, heavy_lo, heavy_hi);
  else
    hipLaunchKernelGGL(fill_kernel<float>, dim3(grid), dim3(256), 0, stream, (float*)ll, n_obs, n_draws,
                       row0, seed, k_lo, k_hi, heavy_lo, heavy_hi);
  return hipGetLastError();
}

}  // namespace pla
