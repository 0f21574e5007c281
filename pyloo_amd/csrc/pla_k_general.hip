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

#include "pla_rows.h"
#include "pla_wave.h"
#include "pla_chunked.h"
#include "pla_launch.h"

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
constexpr int kRedChunks = 512;
constexpr int kRedSlots = 8;  // n, sum loo, M2, sum lppd, #high, #non-finite, min diag, unused

__device__ __forceinline__ void reduce_chunk(const ReduceParams& P, double* part) {
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

__global__ __launch_bounds__(kRedBlock) void reduce_stage1(ReduceParams P, double* part) { reduce_chunk(P, part); }

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
__device__ __forceinline__ void reduce_merge_wave(const ReduceParams& P, const double* part, int nchunks, int lane) {
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
// One wave: lane l merges chunks l*per .. (l+1)*per-1 in order, then the 64 partial results are merged
// by a fixed shuffle tree.  The grouping depends only on the chunk count, so the result is
// reproducible run to run.
__global__ __launch_bounds__(kWave) void reduce_stage2(ReduceParams P, const double* part, int nchunks) {
  reduce_merge_wave(P, part, nchunks, threadIdx.x);
}
// Both stages in ONE launch (a kernel boundary less behind every LOO pass): every workgroup writes its chunk's moments, makes
// them visible at agent scope and takes a ticket; the workgroup whose ticket is the last one merges all chunks -- in chunk
// order, whichever workgroup it is, so the bits do not depend on who came last -- and puts the ticket counter back to zero.
// (`ticket`: one unsigned behind the partials, zero before the first launch and after every launch)
__global__ __launch_bounds__(kRedBlock) void reduce_fused(ReduceParams P, double* part, unsigned* ticket) {
  __shared__ unsigned s_last;
  reduce_chunk(P, part);
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this chunk's moments before the ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the write-back has left before the ticket is taken: MI355X_MICROARCH.md, compiler hazard)
    s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  if (threadIdx.x < kWave) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every other chunk's moments behind their tickets
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    reduce_merge_wave(P, part, (int)gridDim.x, threadIdx.x);
    if (threadIdx.x == 0) *ticket = 0u;
  }
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
void note_kernels(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_kernels, sizeof(g_last_kernels), fmt, ap);
  va_end(ap);
}
const char* last_rows_kernels() { return g_last_kernels; }
template <typename T> static const char* tname() { return sizeof(T) == 8 ? "double" : "float"; }

size_t general_smem_bytes(const RowsParams& p) { return smem_bytes(256, p.tail_cap); }

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
template <typename T, bool LW>
static hipError_t launch_general_t(const RowsParams& p, hipStream_t stream) {
  constexpr int BLOCK = 256;
  if (p.stride_draw == 1 && p.n_draws <= BLOCK * 16 && p.n_draws > BLOCK * 4 && env_flag("PLA_FORCE_PATH") != 1)
    return launch_one<T, BLOCK, 16, LW>(p, stream);
  return launch_one<T, BLOCK, 0, LW>(p, stream);
}
hipError_t launch_general(const RowsParams& p, int dtype, bool lw, hipStream_t stream) {
  if (dtype == PLA_F64) return lw ? launch_general_t<double, true>(p, stream) : launch_general_t<double, false>(p, stream);
  return lw ? launch_general_t<float, true>(p, stream) : launch_general_t<float, false>(p, stream);
}
// general kernel over whatever a fast path declined (usually nothing).  block 1024: the rows are few, so what they cost is the
// LATENCY of one workgroup walking one long row thirteen times -- 1024 threads per row instead of 256 for rows of 8192+ draws
template <typename T, bool LW>
static hipError_t launch_slow_t(const RowsParams& p, hipStream_t stream, int block) {
  const int64_t g = p.n_obs < 1024 ? p.n_obs : 1024;
  if (block == 1024)
    hipLaunchKernelGGL((slow_rows_kernel<T, 1024, LW>), dim3((unsigned)g), dim3(1024), smem_bytes(1024, p.tail_cap), stream, p);
  else
    hipLaunchKernelGGL((slow_rows_kernel<T, 256, LW>), dim3((unsigned)g), dim3(256), smem_bytes(256, p.tail_cap), stream, p);
  return hipGetLastError();
}
hipError_t launch_slow_rows(const RowsParams& p, int dtype, bool lw, hipStream_t stream, int block) {
  if (dtype == PLA_F64) return lw ? launch_slow_t<double, true>(p, stream, block) : launch_slow_t<double, false>(p, stream, block);
  return lw ? launch_slow_t<float, true>(p, stream, block) : launch_slow_t<float, false>(p, stream, block);
}

// Speculative threshold of the wave kernel: the kq-th smallest (roughly) of the 64 per-lane maxima
// over the first gsz register slots.  For exchangeable draws a draw lies below it with probability F,
// F^gsz = kq/64.  Pick (gsz, kq) so that ~2.2(M+1) draws lie above, with kq large enough for the order
// statistic to be stable.  Returns false when no setting fits (the general kernel takes the call).
// `cand_cap`: capacity of the LDS candidate list; only the first min(S, 4096) draws feed the maxima.
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
  static const int off = exp_flag("PLA_NO_THRESHOLD_CHECK");  // (experiment builds only)
  chk->cr_lo = (int)std::ceil(1.25 * (M + 1) / S * S0);
  chk->cr_hi = off ? 0 : (int)std::floor(0.85 * cand_cap / S * S0);
  return true;
}

// Workgroups of a wave-per-row kernel with `waves` waves per workgroup: 512 workgroups are resident (2 per CU), so the grid is
// 512 x 2^k -- whole rounds -- with k as large as leaves every wave >= 24 rows (a wave's first row is loaded without overlap:
// short-lived waves pay that start-up again and again), at most 8 rounds (many rounds even out clock and memory-channel luck)
int64_t wave_grid(int64_t n_obs, int waves) {
  const int64_t need = (n_obs + waves - 1) / waves;
  if (need <= 512) return need < 1 ? 1 : need;
  int64_t grid = 512;
  while (grid < 4096 && n_obs / (2 * grid * waves) >= 24) grid *= 2;
  return grid;
}

// which kernels take a call: the one-chunk wave kernel, the SIS / TIS kernel, the chunked wave kernel, or the general kernel
template <typename T>
static hipError_t launch_typed(const RowsParams& p, const bool LW, hipStream_t stream, hipEvent_t after_first, bool* recorded,
                               const PipeStreams* pipe = nullptr, int* plan = nullptr) {
  if (plan) *plan = 0;
  constexpr int BLOCK = 256;
  constexpr int dtype = sizeof(T) == 8 ? PLA_F64 : PLA_F32;
  const bool unit = (p.stride_draw == 1);
  const int path = env_flag("PLA_FORCE_PATH");  // 0 auto, 1 general kernel only (tests; read per call)
  constexpr int WVEC = 16 / sizeof(T);
  bool waligned = ((uintptr_t)p.in % 16 == 0) && (p.stride_obs % WVEC == 0) && (p.n_draws % WVEC == 0);
  if (LW) waligned = waligned && ((uintptr_t)p.lw_out % 16 == 0);  // rows of lw_out are n_draws apart
  if (path != 1 && unit && waligned && p.method == PLA_PSIS && p.slow_list && p.counters && p.l1_table &&
      p.n_draws <= kWave * kWaveSlots && p.n_draws >= 256 && p.tail_count <= kWaveMaxTail &&
      smem_bytes(BLOCK, p.tail_cap) <= 64 * 1024 && p.n_obs <= 0xffffffffll) {
    int gsz = 0, kq = 0, bits = 0;
    ThresholdCheck chk{};
    if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk))
      return launch_wave_t<T>(p, LW, gsz, kq, bits, chk, stream, after_first, recorded, pipe, plan);
  }
  if (path != 1 && unit && waligned && (p.method == PLA_SIS || p.method == PLA_TIS) && p.slow_list && p.counters &&
      p.n_draws <= kWave * kWaveSlots && p.n_draws >= kWave * WVEC && p.n_obs <= 0xffffffffll)
    return plan ? hipSuccess : launch_is_t<T>(p, LW, stream);
  {
    // rows beyond one register chunk or tails beyond the small kernel's LDS: the chunked kernel
    // (weights mode: the candidates carry 16-bit draw indices, so rows up to 65 536 draws; two LDS capacities)
    const int last_chunk = p.n_draws - ((p.n_draws - 1) / kChunkDraws) * kChunkDraws;
    if (path != 1 && unit && waligned && p.method == PLA_PSIS && p.slow_list && p.counters && p.l1_table &&
        p.n_draws >= 256 && p.n_draws <= (1 << 20) && last_chunk >= kWave * WVEC && p.tail_count <= CapsBig::kMaxTail &&
        (!LW || p.n_draws <= 65536) && smem_bytes(BLOCK, p.tail_cap) <= 64 * 1024 && p.n_obs <= 0xffffffffll) {
      int gsz = 0, kq = 0, bits = 0;
      ThresholdCheck chk{};
      const auto chunked = [&](int caps) { return plan ? hipSuccess : launch_chunked_t<T>(p, LW, caps, gsz, kq, bits, chk, stream, after_first, recorded); };
      if (LW) {
        // (f32 rows: six waves per CU; f64 rows need more than 256 registers per lane next to the row, so four)
        using CapLW = std::conditional_t<sizeof(T) == 4, CapsMidLW, CapsMid>;
        if (p.tail_count <= CapLW::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapLW::kCand))
          return chunked(sizeof(T) == 4 ? kCapsMidLW : kCapsMid);
        if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsBig::kCand)) return chunked(kCapsBig);
      } else {
        if (p.tail_count <= CapsMid4::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsMid4::kCand))
          return chunked(kCapsMid4);
        if (p.tail_count <= CapsMid::kMaxTail && wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsMid::kCand))
          return chunked(kCapsMid);
        if (wave_threshold_params(p.n_draws, WVEC, p.tail_count, &gsz, &kq, &bits, &chk, CapsBig::kCand)) return chunked(kCapsBig);
      }
    }
  }
  if (plan) return hipSuccess;
  return launch_general(p, dtype, LW, stream);
}

hipError_t launch_rows(const RowsParams& p, int dtype, bool lw_mode, hipStream_t stream, hipEvent_t after_first, bool* recorded,
                       const PipeStreams* pipe) {
  if (p.n_obs <= 0) return hipSuccess;
  if (dtype == PLA_F64) return launch_typed<double>(p, lw_mode, stream, after_first, recorded, pipe);
  return launch_typed<float>(p, lw_mode, stream, after_first, recorded, pipe);
}

bool rows_stream_planned(const RowsParams& p, int dtype) {
  if (p.n_obs <= 0) return false;
  int plan = 0;
  PipeStreams probe{};
  probe.sync = reinterpret_cast<unsigned*>(p.counters);  // (any non-null pointer: looked at, not dereferenced)
  if (dtype == PLA_F64) (void)launch_typed<double>(p, false, nullptr, nullptr, nullptr, &probe, &plan);
  else (void)launch_typed<float>(p, false, nullptr, nullptr, nullptr, &probe, &plan);
  return plan != 0;
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
int reduce_workspace_doubles() { return kRedChunks * kRedSlots + 2; }  // (+ the ticket counter of reduce_fused: zeroed by the engine)

#ifndef PLA_REDUCE_FUSED
#define PLA_REDUCE_FUSED 1
#endif
hipError_t launch_reduce(const ReduceParams& p, double* workspace, hipStream_t stream) {
  int64_t chunks = (p.n_obs + 1023) / 1024;  // >= 1024 observations per chunk, at most kRedChunks of them
  if (chunks < 1) chunks = 1;
  if (chunks > kRedChunks) chunks = kRedChunks;
#if PLA_REDUCE_FUSED
  hipLaunchKernelGGL(reduce_fused, dim3((unsigned)chunks), dim3(kRedBlock), 0, stream, p, workspace,
                     reinterpret_cast<unsigned*>(workspace + kRedChunks * kRedSlots));
  return hipGetLastError();
#else
  hipLaunchKernelGGL(reduce_stage1, dim3((unsigned)chunks), dim3(kRedBlock), 0, stream, p, workspace);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(kWave), 0, stream, p, workspace, (int)chunks);
  return hipGetLastError();
#endif
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
