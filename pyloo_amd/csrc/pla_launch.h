// Host-side pieces shared by the translation units of libpyloo_amd.so (internal).  The kernels are compiled as several
// units in parallel (pyloo_amd/build.py): pla_k_general.hip (general kernel, reductions, dispatcher), pla_k_wave_f64/f32.hip,
// pla_k_chunked_f64/f32.hip, pla_k_fit.hip, pla_k_waic.hip, pla_k_col.hip, pla_k_eloo.hip; each launches the kernels it defines.
#pragma once

#include <cstdlib>

#include "../../include/pyloo_amd.h"
#include "pla_fast.h"
#include "pla_kernels.h"

namespace pla {

// ---- run-time switches -----------------------------------------------------------------------------------------------------
// The shipped library reads exactly these environment variables, all of them PATH SELECTORS: every setting computes the
// reference's results, through another arrangement of the same kernels (the tests compare the arrangements with one another):
//   PLA_PIPE=0               split LOO pass back to back on the caller's stream instead of streamed (fit kernel beside the sweep)
//   PLA_STREAM_PATIENCE_US   how long the streamed fit kernel waits for a chunk nobody produces before it leaves the rest to the
//                            plain fit kernel (a test sets 1)
//   PLA_FORCE_PATH=1         general kernel only
//   PLA_INGEST_TRANSPOSE=1   observations-fastest device matrices through the transposing ingestion instead of being read in place
//   PLA_INGEST_BLOCK_MB      block size of that ingestion
// pla_env_overrides() (C ABI) names the ones that are set, so that a benchmark record can say which arrangement ran.
// Everything else -- phase ablation, grids, priorities, switching checks off: knobs that change timings or even results -- exists
// only in builds with -DPLA_EXPERIMENT (tools/build_alt.sh) and is compiled out of the default library.
inline int env_flag(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
#if defined(PLA_EXPERIMENT)
inline int exp_flag(const char* name) { return env_flag(name); }
inline const char* exp_str(const char* name) { return getenv(name); }
constexpr bool kExperiment = true;
#else
constexpr int exp_flag(const char*) { return 0; }
constexpr const char* exp_str(const char*) { return nullptr; }
constexpr bool kExperiment = false;
#endif

// ---- what the launchers share ----------------------------------------------------------------------------------------------
struct ThresholdCheck {  // FastParams::cr_lo, cr_hi
  int cr_lo, cr_hi;
};
inline int isqrt_host(int n) {
  int r = 0;
  while ((r + 1) * (r + 1) <= n) ++r;
  return r;
}
inline int mest_for(int tail_count) { return 30 + isqrt_host(tail_count); }  // psis.py:184

// text for pla_engine_last_kernels: what the last launch_rows() of this thread launched
void note_kernels(const char* fmt, ...);
inline const char* dtype_name(int dtype) { return dtype == PLA_F64 ? "double" : "float"; }

// general kernel (pla_rows.h): every row / the rows a fast path declined (device list)
hipError_t launch_general(const RowsParams& p, int dtype, bool lw, hipStream_t stream);
hipError_t launch_slow_rows(const RowsParams& p, int dtype, bool lw, hipStream_t stream, int block = 256);
// output pass of the split weights pass for long rows (pla_lwout.h): behind the selection and the fit kernel
hipError_t launch_lw_output(const RowsParams& p, int dtype, hipStream_t stream);
size_t general_smem_bytes(const RowsParams& p);

// fit kernels of the split pass (pla_fit.h)
bool split_ok(const RowsParams& p, int mestM);
hipError_t launch_fit(const RowsParams& p, const FastParams& f, int mestM, hipStream_t stream, const unsigned* fitted = nullptr,
                      const unsigned* gave_up = nullptr);
hipError_t launch_fit_stream(const RowsParams& p, const FastParams& f, int mestM, unsigned* sync, hipStream_t stream, bool helper = false);
// zeroes the counters and the flags of a streamed pass in one launch
hipError_t launch_zero_sync(unsigned long long* counters, bool all_counters, unsigned* sync, int64_t n_obs, hipStream_t stream);
// layout of PipeStreams::sync (unsigned words): [0] row queue of the first kernel, [16] chunk queue of the fit kernel,
// [32] "the fit kernel gave up waiting", [48 ..) one flag per chunk, then one "fitted" flag per chunk
constexpr int kSyncQueue = 0, kSyncTake = 16, kSyncGaveUp = 32, kSyncDone = 48;
// RowsParams::counters (unsigned long long, device): [0 .. kCountersPerCall) belong to one call and are zeroed by it ([0] rows on
// the slow list, [1] running total, [2..3] clock probe, [4] group counter of the tile kernel, [8..15] reasons in profiling builds);
// from kCountersPerCall on they live as long as the engine: [kCounterGaveUp] passes in which the streamed fit kernel gave up
constexpr int kCountersPerCall = 16, kCounterGaveUp = 16, kCountersTotal = 32;

// wave-per-observation kernels (pla_wave.h, pla_chunked.h, pla_is.h): one launcher per input dtype, each in a unit of its own
int64_t wave_grid(int64_t n_obs, int waves);
template <typename T>
hipError_t launch_wave_t(const RowsParams& p, bool lw, int gsz, int kq, int bits, const ThresholdCheck& chk, hipStream_t stream,
                         hipEvent_t after_first, bool* recorded, const PipeStreams* pipe, int* plan_stream);
template <typename T>
hipError_t launch_is_t(const RowsParams& p, bool lw, hipStream_t stream);
enum ChunkedCaps { kCapsMid4 = 0, kCapsMid = 1, kCapsBig = 2, kCapsMidLW = 3 };
template <typename T>
hipError_t launch_chunked_t(const RowsParams& p, bool lw, int caps, int gsz, int kq, int bits, const ThresholdCheck& chk,
                            hipStream_t stream, hipEvent_t after_first, bool* recorded);

}  // namespace pla
