// Host-visible launch interface of the HIP kernels (internal to libpyloo_amd.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pla {

struct RowsParams {
  const void* in;       // (n_obs, n_draws) log-likelihood (LOO mode) or log ratios (LW mode)
  int64_t n_obs;
  int n_draws;
  int64_t stride_obs;   // elements
  int64_t stride_draw;  // elements
  int method;           // PLA_PSIS / PLA_SIS / PLA_TIS
  int tail_count;       // M (PSIS)
  int tail_cap;         // power of two >= M: LDS tail capacity
  double scale_value;   // LOO mode
  double* diag;         // [n_obs] or null
  double* loo_i;        // [n_obs] or null (LOO mode)
  double* lppd_i;       // [n_obs] or null (LOO mode)
  void* lw_out;         // (n_obs, n_draws) contiguous, input dtype (LW mode)
  unsigned long long* counters;  // [4] device counters: [0] rows left to the general kernel
  unsigned* slow_list;           // [n_obs] workspace for the fast path (may be null: general kernel only)
  const double* l1_table;        // [tail_count] log1p(-(j+0.5)/M), then [64] 1 - sqrt(m_est/(j+0.5)) (host-computed)
  // hand-over buffers of the split LOO pass (wave kernel -> fit kernel, pla_fit.h); null: fused pass
  double* ws_y = nullptr;        // [n_obs][ws_stride]
  double* ws_s = nullptr;        // [n_obs][8]
  int ws_stride = 0;
  // optional row selection (loo_subsample, loo_subsample.py:316-330): observation r of this call is row row_index[r] of `in`;
  // outputs stay compact ([n_obs] = number of selected rows)
  const int64_t* row_index = nullptr;  // (already clamped into the matrix: launch_clamp_rows)
  int ws_sstride = 8;            // doubles per observation in ws_s (16 in the streamed pass: one 128-byte line each)
  // weights mode of the split pass (rows longer than the registers: selection kernel -> fit kernel -> lw_output_kernel,
  // pla_lwout.h): the hand-over buffers above are there for it; false: the fused weights kernels
  bool lw_split = false;
};

// Streamed split pass: the first kernel (statistics, sweep, selection; HBM stream) on `first` and, BESIDE it on `second`, the
// fit kernel, which takes the chunks of observations as the first kernel finishes them (FastParams::done, pla_fit.h).  Both
// streams are forked from and joined to the caller's inside launch_rows(); `sync` is device memory for the flags and queues of
// one launch (stream_sync_bytes(n_obs)), zeroed by the launcher.
struct PipeStreams {
  hipStream_t first;
  hipStream_t second;
  hipEvent_t fork, join_first, join_second;  // (timing disabled)
  unsigned* sync;
  hipEvent_t before_first;  // optional (timing): recorded on `first` right before the first kernel ...
  hipEvent_t after_first;   // ... and right behind it
  bool zero_all_counters;   // first launch of a call: RowsParams::counters[0..15] are zeroed with the flags (else [0] only)
};
size_t stream_sync_bytes(int64_t n_obs);

// element offset of observation r's row in the input matrix
#define PLA_ROW_OFFSET(P, r) (((P).row_index ? (P).row_index[(r)] : (int64_t)(r)) * (P).stride_obs)

struct ReduceParams {
  const double* diag;
  const double* loo_i;
  const double* lppd_i;
  int64_t n_obs;
  double good_k;
  double* agg;  // [PLA_AGG_COUNT]
  const unsigned long long* counters;  // [0] -> PLA_AGG_N_SLOW (may be null)
};

// returns hipSuccess or the launch error; never synchronises
// `after_first`: optional event recorded right after the first kernel of a split LOO pass (timing of the dominant kernel alone);
// *recorded is set when it was
hipError_t launch_rows(const RowsParams& p, int dtype, bool lw_mode, hipStream_t stream, hipEvent_t after_first = nullptr,
                       bool* recorded = nullptr, const PipeStreams* pipe = nullptr);
// true when launch_rows() would run these rows as a split pass (first kernel + fit kernel): the shapes the pipeline serves
// true when launch_rows() given a PipeStreams would run these rows as a streamed split pass
bool rows_stream_planned(const RowsParams& p, int dtype);
hipError_t launch_reduce(const ReduceParams& p, double* workspace, hipStream_t stream);
// out[i] = min(max(in[i], 0), n_src - 1): a caller's device index list can never make a row kernel read outside the matrix
hipError_t launch_clamp_rows(const int64_t* in, int64_t n_rows, int64_t n_src, int64_t* out, hipStream_t stream);
int reduce_workspace_doubles();  // size of `workspace` (device memory)
// observation-sharded runs: a rank's row of the world x 8 table (the other rows zero) / the Chan merge of the summed table
hipError_t launch_aggregate_pack(const double* agg, int rank, int world, double* table, hipStream_t stream);
hipError_t launch_aggregate_merge(const double* table, int world, double* out, hipStream_t stream);
// text for pla_engine_last_kernels: what the last launch_rows() on this thread launched
const char* last_rows_kernels();
// Observations-fastest ingestion: ArviZ keeps log-likelihoods as (chain, draw, *obs), so the
// (obs, sample) view pyloo stacks (loo.py:189) has unit stride along the observations.  Rows [obs0, obs0 + n_rows) of
// such a matrix (element (i, s) at in[s * stride_draw + i]) are written as a contiguous (n_rows, n_draws) block.
hipError_t launch_transpose_rows(const void* in, int dtype, int64_t stride_draw, int64_t obs0, int64_t n_rows, int n_draws,
                                 void* out, hipStream_t stream);
hipError_t launch_fill_chains(void* ll, int dtype, int64_t n_obs, int64_t n_draws, int chains, double rho, double off_sd, int64_t row0,
                              uint64_t seed, double k_lo, double k_hi, hipStream_t stream);
hipError_t launch_fill_synthetic(void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t row0,
                                 uint64_t seed, double k_lo, double k_hi, double heavy_lo,
                                 double heavy_hi, hipStream_t stream);
// WAIC pass (waic.py:109-160): lppd_i, var_i, waic_i per observation; `replaced` counts NaN/inf entries
hipError_t launch_waic(const void* in, const int64_t* row_index, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs, int64_t stride_draw,
                       double scale_value, double* lppd_i, double* var_i, double* waic_i,
                       unsigned long long* replaced, hipStream_t stream);
// weighted expectations + function-specific Pareto k (e_loo.py:56-264): x, lw, lr share dtype, shape and strides; lr may equal lw
hipError_t launch_e_loo(const void* x, const void* lw, const void* lr, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs,
                        int64_t stride_draw, int tail_len, double* mean, double* var, double* k_mean, double* k_var, double* k_none,
                        unsigned* slow_list, unsigned long long* slow_count,
                        hipStream_t stream);
// weighted quantiles of e_loo (e_loo.py:468-515, 534-554): out[n_obs][n_probs]; probs is a DEVICE pointer
hipError_t launch_e_loo_quantiles(const void* x, const void* lw, int dtype, int64_t n_obs, int n_draws, int64_t stride_obs,
                                  int64_t stride_draw, const double* probs, int n_probs, double* out, unsigned* slow_list,
                                  unsigned long long* slow_count, hipStream_t stream);  // slow_list: [n_obs] scratch (may be null)
// Observations-fastest PSIS-LOO without a transposing pass (pla_col.h): lane-per-observation sweep + per-observation
// selection + the fit kernel of the split pass.  `p`: in = first observation of the block, stride_obs = 1, stride_draw = ld,
// ws_y / ws_s / slow_list / counters / l1_table set as for the split pass.  col_ws: col_workspace_bytes(n_obs) of device memory.
bool col_supported(int n_draws, int tail_count, int* kq);
size_t col_workspace_bytes(int64_t n_obs);
hipError_t launch_col(const RowsParams& p, int dtype, int kq, void* col_ws, hipStream_t stream);
// ... a workgroup per 16 observations, candidate lists in LDS (pla_tile.h): f64 matrices, 512 <= n_draws, tail counts <= 250 and a
// list capacity that leaves room on both sides of the expected candidate count.  No workspace beyond the split pass's hand-over.
// streamed (pipe with its flags, ws_sstride 16): the fit kernel runs beside the tile kernel, as in the streamed pass of the row kernels;
// the lists are shorter then (the fit kernel's workgroup needs its share of the CU's LDS), so `streamed` goes into the shape test.
bool tile_supported(int dtype, int n_draws, int tail_count, int64_t ld, bool streamed, int* ks);
hipError_t launch_tile(const RowsParams& p, int dtype, int ks, hipStream_t stream, const PipeStreams* pipe = nullptr);
// WAIC on an observations-fastest matrix read in place, one lane per observation (element (i, s) at in[s * ld + i])
hipError_t launch_waic_col(const void* in, int dtype, int64_t n_obs, int n_draws, int64_t ld, double scale_value, double* lppd_i,
                           double* var_i, double* waic_i, unsigned long long* replaced, hipStream_t stream);
// largest tail count the kernels accept
int max_tail_count();

}  // namespace pla
