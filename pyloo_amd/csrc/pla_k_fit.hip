// launchers of the fit kernels (pla_fit.h)
// (one translation unit of libpyloo_amd.so: the kernels are compiled in parallel, pyloo_amd/build.py)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pla_fit.h"
#include "pla_launch.h"

namespace pla {

size_t stream_sync_bytes(int64_t n_obs);
// second kernel of a split LOO pass: fit / smoothing / outputs for the tails the selection kernel handed over (pla_fit.h)
hipError_t launch_fit(const RowsParams& p, const FastParams& f, int mestM, hipStream_t stream, const unsigned* fitted, const unsigned* gave_up) {
  FitParams q{p.ws_y, p.ws_s, p.ws_stride, p.n_obs, p.n_draws, p.tail_count, mestM, f.log_S, p.scale_value, p.l1_table,
              p.l1_table + p.tail_count, p.diag, p.loo_i, p.lppd_i, p.slow_list, p.counters};
  q.slow_base = f.slow_base;
  q.ws_sstride = f.ws_sstride;
  q.fitted = const_cast<unsigned*>(fitted);  // (behind a streamed pass: only the chunks that pass left)
  q.gave_up = const_cast<unsigned*>(gave_up);
  q.gave_up_total = p.counters + kCounterGaveUp;
  q.lw_mode = p.lw_split;  // (weights mode of the split pass: pla_lwout.h)
  static const int skip_fit = exp_flag("PLA_SKIP_FIT");  // timing experiments only (experiment builds): the outputs are then garbage
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
hipError_t launch_fit_stream(const RowsParams& p, const FastParams& f, int mestM, unsigned* sync, hipStream_t stream, bool helper) {
  FitParams q{p.ws_y, p.ws_s, p.ws_stride, p.n_obs, p.n_draws, p.tail_count, mestM, f.log_S, p.scale_value, p.l1_table,
              p.l1_table + p.tail_count, p.diag, p.loo_i, p.lppd_i, p.slow_list, p.counters};
  q.slow_base = f.slow_base;
  q.ws_sstride = f.ws_sstride;
  const int64_t nchunks = (p.n_obs + kQueueChunk - 1) / kQueueChunk;
  q.take = sync + kSyncTake;
  q.gave_up = sync + kSyncGaveUp;
  q.done = sync + kSyncDone;
  q.fitted = sync + kSyncDone + nchunks;
  q.producer = sync + kSyncQueue;
  // how long a chunk's flag may stay down while the producer's row queue stands still, in ticks of the 100 MHz real-time
  // counter: 20 ms once the producer has taken rows, 2 ms while it has not taken one (it was launched first).  Then the rest is
  // left to the plain fit kernel that follows -- nothing depends on the two kernels having been run side by side, and a
  // profiler that serialises them costs milliseconds, not seconds.  PLA_STREAM_PATIENCE_US: both bounds (a test sets 1).
  const int patience_us = env_flag("PLA_STREAM_PATIENCE_US");  // (read per call)
  q.patience = patience_us > 0 ? (unsigned)patience_us * 100u : 2000000u;
  q.patience_start = patience_us > 0 ? (unsigned)patience_us * 100u : 200000u;
  static const int fg_forced = exp_flag("PLA_FIT_GRID");  // (experiment builds only)
  int64_t g = nchunks < 256 ? nchunks : 256;
  if (fg_forced > 0 && fg_forced < nchunks) g = fg_forced;
  if (helper) {
    // a second launch of the same kernel BEHIND the wave kernel on its stream: the fit kernel beside the wave kernel lives on
    // what that kernel leaves it and ends a few per cent of the chunks behind; once the wave kernel is gone these workgroups
    // (three more per CU) take chunks from the same counter and the tail is over in a few trips
    static const int hg = exp_flag("PLA_FIT_HELPERS");  // (experiment builds only)
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
hipError_t launch_zero_sync(unsigned long long* counters, bool all_counters, unsigned* sync, int64_t n_obs, hipStream_t stream) {
  const size_t nsync = stream_sync_bytes(n_obs) / sizeof(unsigned);
  const unsigned zg = (unsigned)((nsync + 1023) / 1024 < 256 ? (nsync + 1023) / 1024 : 256);
  hipLaunchKernelGGL(zero2_kernel, dim3(zg ? zg : 1), dim3(256), 0, stream, reinterpret_cast<unsigned*>(counters),
                     (size_t)(all_counters ? 2 * kCountersPerCall : 2), sync, nsync);
  return hipGetLastError();
}
size_t stream_sync_bytes(int64_t n_obs) {
  const int64_t nchunks = (n_obs + kQueueChunk - 1) / kQueueChunk;
  return (size_t)(kSyncDone + 2 * nchunks + 16) * sizeof(unsigned);
}
// shapes the split pass covers: hand-over buffers present, tail within the stride, grid within the fit kernel's lanes
bool split_ok(const RowsParams& p, int mestM) {
  if (!p.ws_y || !p.ws_s || p.ws_stride % 64 != 0 || p.tail_count > p.ws_stride) return false;
  return p.ws_stride <= 256 ? mestM <= kFitGrid : (p.ws_stride <= 448 && mestM <= kFitGridBig);
}

}  // namespace pla
