// launcher of the output pass of the split weights pass (pla_lwout.h)
// (one translation unit of libpyloo_amd.so: the kernels are compiled in parallel, pyloo_amd/build.py)
#include "pla_lwout.h"
#include "pla_launch.h"

namespace pla {

hipError_t launch_lw_output(const RowsParams& p, int dtype, hipStream_t stream) {
  if (p.n_obs <= 0) return hipSuccess;
  if (!p.ws_y || !p.lw_split || !p.ws_s || !p.lw_out || !p.l1_table || p.ws_sstride < 8 || p.stride_draw != 1 || p.row_index || p.tail_count > kLwoTail || p.ws_stride < p.tail_count)
    return hipErrorInvalidValue;
  LwOutParams q{p.in, p.lw_out, p.n_obs, p.n_draws, p.stride_obs, p.ws_y, p.ws_s, p.ws_stride, p.ws_sstride, p.l1_table, p.tail_count};
  int64_t grid = (p.n_obs + kLwoWaves - 1) / kLwoWaves;
  if (grid > 768) grid = 768;  // (three workgroups per CU, resident for the whole launch: the waves walk the rows with a stride)
  if (dtype == PLA_F64) hipLaunchKernelGGL(lw_output_kernel<double>, dim3((unsigned)grid), dim3(kWave * kLwoWaves), 0, stream, q);
  else hipLaunchKernelGGL(lw_output_kernel<float>, dim3((unsigned)grid), dim3(kWave * kLwoWaves), 0, stream, q);
  return hipGetLastError();
}

}  // namespace pla
