// Pieces shared by the fast-path kernels: reduction operators, launch-time parameters.
#pragma once

#include "pla_rows.h"

namespace pla {

// ---- reduction operators ------------------------------------------------------------------
enum RedOp { R_SUM, R_MAX, R_MIN };
template <RedOp OP>
__device__ __forceinline__ double red_apply(double a, double b) {
  if constexpr (OP == R_SUM) return a + b;
  else if constexpr (OP == R_MAX) return fmax(a, b);
  else return fmin(a, b);
}
struct FastParams {
  int gsz;                        // register slots per lane whose maximum feeds the speculative threshold
  int kq;                         // the threshold has >= kq of the 64 per-lane maxima below it
  unsigned* slow_list;            // [n_obs] rows for the general kernel
  unsigned long long* counters;   // [0] = number of rows in slow_list
  int debug_skip;                 // phase-ablation bits, honoured only by PLA_WAVE_ABLATE builds
  const double* l1_table;         // [M] log1p(-(j+0.5)/M), host-computed (wave kernel)
  double log_S;                   // log(n_draws)
  const double* b_grid;           // [64] 1 - sqrt(m_est/(j+0.5)) for m_est = mest_M (psis.py:186)
  int mest_M;                     // 30 + isqrt(M)
  // split pass (pla_fit.h): hand-over buffers of the wave kernel, null when the pass is fused
  double* ws_y = nullptr;         // [n_obs][ws_stride] ascending tail values
  double* ws_s = nullptr;         // [n_obs][8] scalars
  int ws_stride = 0;
};

}  // namespace pla
