// launchers of the observations-fastest LOO kernels (pla_col.h, pla_tile.h)
// (one translation unit of libpyloo_amd.so: the kernels are compiled in parallel, pyloo_amd/build.py)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pla_col.h"
#include "pla_tile.h"
#include "pla_launch.h"

namespace pla {

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
  const int mestM = mest_for(p.tail_count);
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
  return launch_slow_rows(p, dtype, false, stream);
}

// ---- observations-fastest LOO, a workgroup per 16 observations (pla_tile.h) -------------------------------------------------
// The threshold is the sample's ks-th largest value: ks / 512 of the row is expected at or above it, T = ks S / 512 draws, with a
// standard deviation of sqrt(S^2 p (1 - p) / 512 + S p (1 - p)), p = T / S (the sample's quantile, then the row's count given the
// quantile).  T sits halfway between the M + 1 the selection needs and the list's capacity; shapes where that leaves less than
// 3.0 standard deviations on either side stay with the lane-per-observation kernels (pla_col.h).
bool tile_supported(int dtype, int n_draws, int tail_count, int64_t ld, bool streamed, int* ks) {
  static const int off = exp_flag("PLA_NO_TILE");  // (experiment builds only)
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
  const int mestM = mest_for(p.tail_count);
  TileParams c{p.in, p.n_obs, p.n_draws, p.stride_draw, ks};
  FastParams f{0, 0, p.slow_list, p.counters, 0, p.l1_table, std::log((double)p.n_draws), p.l1_table + p.tail_count, mestM};
  f.ws_y = p.ws_y;
  f.ws_s = p.ws_s;
  f.ws_stride = p.ws_stride;
  f.ws_sstride = p.ws_sstride;
  const int64_t ngroups = (p.n_obs + 15) / 16;
  static const int grid_env = exp_flag("PLA_TILE_GRID");  // (experiment builds only)
  const int64_t cap = grid_env > 0 ? grid_env : 256;  // one workgroup per CU
  const unsigned g1 = (unsigned)(ngroups < cap ? ngroups : cap);
  const bool streamed = pipe && pipe->sync && p.ws_sstride == 16 && p.ws_stride <= 256 && p.n_obs < ((int64_t)1 << 31) && split_ok(p, mestM);
  if (streamed) {
    // streamed: the fit kernel runs beside the tile kernel and takes each group of 16 observations as its flag goes up
    // (launch_wave, streamed branch: the same flags, streams and leftovers)
    if (!tile_lds_attr<true>()) return hipErrorInvalidValue;
    unsigned* const sync = pipe->sync;
    const int64_t nchunks = (p.n_obs + kQueueChunk - 1) / kQueueChunk;
    e = launch_zero_sync(p.counters, pipe->zero_all_counters, sync, p.n_obs, stream);
    if (e == hipSuccess) e = hipEventRecord(pipe->fork, stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(pipe->first, pipe->fork, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(pipe->second, pipe->fork, 0);
    if (e != hipSuccess) return e;
    f.queue = sync + kSyncQueue;  // (groups beyond a workgroup's first: zero at launch)
    f.done = sync + kSyncDone;
    static const char* wprio = exp_str("PLA_WAVE_PRIO");  // (experiment builds only)
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
    e = launch_fit(p, f, mestM, stream, sync + kSyncDone + nchunks, sync + kSyncGaveUp);
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
  return launch_slow_rows(p, PLA_F64, false, stream);
}

}  // namespace pla
