// launchers of the e_loo kernels (pla_eloo.h)
// (one translation unit of libpyloo_amd.so: the kernels are compiled in parallel, pyloo_amd/build.py)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pla_eloo.h"
#include "pla_launch.h"

namespace pla {

hipError_t launch_e_loo(const void* x, const void* lw, const void* lr, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs,
                        int64_t stride_draw, int tail_len, double* mean, double* var, double* k_mean, double* k_var, double* k_none,
                        unsigned* slow_list, unsigned long long* slow_count, hipStream_t stream) {
  if (n_obs <= 0) return hipSuccess;
  ELooParams p{x, lw, lr ? lr : lw, n_obs, n_draws, stride_obs, stride_draw, tail_len, mean, var, k_mean, k_var, k_none, slow_list, slow_count};
  const int64_t grid = n_obs < 16384 ? n_obs : 16384;
  const int path = env_flag("PLA_FORCE_PATH");  // 1: general kernel only (tests)
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
  // Rows of up to 4096 contiguous, 16-byte aligned draws: ONE WAVE per observation, no workgroup barriers (e_loo_quantile_wave_kernel);
  // the few rows it declines are listed for the 512-thread kernel behind it, which takes every other shape whole.
  const int vec = dtype == PLA_F64 ? 2 : 4;
  const bool wave = slow_list && slow_count && n_draws <= 4096 && n_obs <= 0xffffffffll && env_flag("PLA_FORCE_PATH") != 1 && stride_draw == 1 &&
                    (((uintptr_t)x | (uintptr_t)lw) % 16 == 0) && stride_obs % vec == 0 && n_draws % vec == 0 && n_draws >= kWave * vec;
  if (wave) {
    hipError_t e = hipMemsetAsync(slow_count, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    p.slow_list = slow_list;
    p.slow_count = slow_count;
    int64_t wg = (n_obs + 3) / 4;
    if (wg > 4096) wg = 4096;
    if (dtype == PLA_F64) hipLaunchKernelGGL(e_loo_quantile_wave_kernel<double>, dim3((unsigned)wg), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(e_loo_quantile_wave_kernel<float>, dim3((unsigned)wg), dim3(256), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int64_t g2 = wave ? (n_obs < 2048 ? n_obs : 2048) : (n_obs < 16384 ? n_obs : 16384);
  if (dtype == PLA_F64) hipLaunchKernelGGL((e_loo_quantile_kernel<double, 512>), dim3((unsigned)g2), dim3(512), 0, stream, p);
  else hipLaunchKernelGGL((e_loo_quantile_kernel<float, 512>), dim3((unsigned)g2), dim3(512), 0, stream, p);
  return hipGetLastError();
}

}  // namespace pla
