// launchers of the WAIC kernels (pla_waic.h)
// (one translation unit of libpyloo_amd.so: the kernels are compiled in parallel, pyloo_amd/build.py)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pla_waic.h"
#include "pla_launch.h"

namespace pla {

template <typename T>
static hipError_t launch_waic_typed(const WaicParams& p, hipStream_t stream) {
  constexpr int WVEC = 16 / sizeof(T);
  const bool fast = p.stride_draw == 1 && ((uintptr_t)p.in % 16 == 0) && (p.stride_obs % WVEC == 0) &&
                    (p.n_draws % WVEC == 0) && p.n_draws <= kWave * kWaveSlots && p.n_draws >= kWave * WVEC;
  const int path = env_flag("PLA_FORCE_PATH");  // 1: general kernel only (tests)
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
hipError_t launch_waic(const void* in, const int64_t* row_index, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs, int64_t stride_draw,
                       double scale_value, double* lppd_i, double* var_i, double* waic_i,
                       unsigned long long* replaced, hipStream_t stream) {
  if (n_obs <= 0) return hipSuccess;
  WaicParams p{in, n_obs, n_draws, stride_obs, stride_draw, scale_value, lppd_i, var_i, waic_i, replaced, row_index};
  return dtype == PLA_F64 ? launch_waic_typed<double>(p, stream) : launch_waic_typed<float>(p, stream);
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

}  // namespace pla
