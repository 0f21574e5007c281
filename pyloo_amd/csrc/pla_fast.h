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
// ---- which draws feed the speculative threshold ----------------------------------------------
// The statistics pass visits the 16-byte vectors of a lane in an order whose first 2^B entries are the bit reversal of
// 0 .. 2^B - 1 (then the rest in natural order), and the per-lane maximum is snapshotted after the first `gsz` slots: the
// sample is then spread evenly over the first 2^B vectors of the row -- every chain of a chain-major (chain, draw) stack
// contributes -- instead of being the row's first draws.  B is the largest of {log2 NQ, log2 NQ - 1, log2 NQ - 2} whose
// span is mostly real draws, 0 (natural order) for very short rows.
__host__ __device__ constexpr int bitrev_order(int i, int B) {
  if (i >= (1 << B)) return i;
  int r = 0;
  for (int b = 0; b < B; ++b) r |= ((i >> b) & 1) << (B - 1 - b);
  return r;
}
__host__ __device__ constexpr int sample_bits_for(int qfull, int log_nq) {
#if defined(PLA_SAMPLE_NATURAL)
  return 0;  // A/B knob: the row's first draws, as in round 1
#endif
  for (int b = log_nq; b >= log_nq - 2 && b > 0; --b)
    if (3 * (1 << b) <= 4 * qfull) return b;
  return 0;
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
  int sample_bits = 0;            // B of bitrev_order: which slots the first `gsz` visited ones are
  // check of the speculative threshold before the sweep (wave_threshold_check): it is kept when cr_lo <= #(draws of the
  // register block at or above it) <= cr_hi, otherwise replaced by one found by bisection on such counts.  cr_hi == 0: no check.
  int cr_lo = 0, cr_hi = 0;
  double* ws_y = nullptr;         // [n_obs][ws_stride] ascending tail values
  double* ws_s = nullptr;         // [n_obs][8] scalars
  int ws_stride = 0;
  unsigned slow_base = 0;         // added to the row numbers written to slow_list (pipelined pass: one list for all blocks)
  // Dynamic row queue (null: row r = first + i * waves, fixed at launch): waves take chunks of kQueueChunk consecutive rows from
  // this counter (zero at launch).  Nothing then depends on every workgroup of the grid being resident at once -- a kernel that
  // shares the chip with another one (pipelined pass) loses the rows of a displaced workgroup to its neighbours, not its time.
  unsigned* queue = nullptr;
  int ws_sstride = 8;             // doubles per observation in ws_s (16 = one 128-byte line each: streamed pass)
  // Streamed split pass (pla_fit.h, fit_rows_stream_kernel): the fit kernel runs BESIDE this kernel and takes the chunks of
  // kQueueChunk rows as they are finished: done[c] is set (agent scope) once every hand-over store of chunk c has drained.
  unsigned* done = nullptr;
  int retry_target = 0;           // long rows that come round again (pla_chunked.h, ChunkRetry): the number of draws a later attempt wants above its threshold (0: rows do not return)
  int prio = 0;                   // s_setprio of the wave kernel's waves (streamed pass: the fit kernel beside it takes what is left)
};
#ifndef PLA_QUEUE_CHUNK
#define PLA_QUEUE_CHUNK 16
#endif
constexpr int kQueueChunk = PLA_QUEUE_CHUNK;
// Rows a wave of the wave kernel takes from the queue at a time: kQueueChunk, or a divisor of it -- the last units of a launch
// are what the waves finish at different times, so the smaller the unit the shorter the stretch at the end of the kernel in
// which the chip runs half empty (and the fit kernel beside it waits for the last flags).  done[c] COUNTS the rows of chunk c
// that have been handed over (the fit kernel takes the chunk once all of them are there): a store of kQueueChunk when the unit
// is the chunk, an agent-scope atomic add per unit otherwise.
#ifndef PLA_QUEUE_UNIT
#define PLA_QUEUE_UNIT 16
#endif
constexpr int kQueueUnit = PLA_QUEUE_UNIT;
static_assert(kQueueChunk % kQueueUnit == 0, "a unit never straddles two chunks");

}  // namespace pla
