// C ABI of libpyloo_amd.so (see include/pyloo_amd.h).  Host code only: argument checks,
// host<->device staging for PLA_HOST callers, kernel launches on the caller's stream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/pyloo_amd.h"
#include "pla_launch.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define PLA_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? PLA_ERR_NOMEM : PLA_ERR_HIP, "%s: %s", #call, \
                  hipGetErrorString(e_));                                                 \
  } while (0)

int pow2_at_least(int64_t n) {
  int p = 8;
  while (p < n) p <<= 1;
  return p;
}

}  // namespace

struct pla_engine {
  int device = 0;
  // One engine = one workspace: every entry point that takes an engine holds this lock for the whole call, so
  // concurrent callers (ctypes releases the GIL) are serialised instead of racing on the buffers below.
  std::mutex mu;
  // frozen: the workspace may be referenced by a captured HIP graph; a call that would have to reallocate any of
  // it returns PLA_ERR_FROZEN instead (pla_engine_set_frozen)
  bool frozen = false;
  unsigned long long* counters = nullptr;  // [pla::kCountersTotal] device (layout: pla_launch.h): [0..15] per call, [16..] engine statistics
  double* d_red = nullptr;                 // reduction partials
  // staging for PLA_HOST callers (grown on demand)
  void* d_in = nullptr;
  size_t d_in_bytes = 0;
  void* d_lw = nullptr;
  size_t d_lw_bytes = 0;
  double* d_pw = nullptr;  // 3 * n doubles + agg
  size_t d_pw_elems = 0;
  void* d_slow = nullptr;  // [n] row list of the fast path
  size_t d_slow_bytes = 0;
  // quantile tables log1p(-(j+0.5)/M), one immutable device buffer per tail count M seen so far (never rewritten
  // or freed before the engine is destroyed: enqueued launches and captured graphs keep valid pointers)
  struct L1Table { int64_t M; double* d; };
  std::vector<L1Table> l1_tables;
  void* d_ws = nullptr;    // hand-over buffers of the split LOO pass: [n][stride] tail values + [n][8] scalars
  size_t d_ws_bytes = 0;
  void* d_col = nullptr;   // observations-fastest LOO (pla_col.h): candidate lists + scalars of one block of observations
  size_t d_col_bytes = 0;
  void* d_probs = nullptr;  // quantile levels of pla_e_loo_quantiles
  size_t d_probs_bytes = 0;
  void* d_rows = nullptr;  // clamped copy of a caller's device row-index list
  void* d_slab = nullptr;  // host path, observations-fastest input: (n_draws, block of observations) slab before the transpose
  size_t d_slab_bytes = 0;
  size_t d_rows_bytes = 0;
  // timing of the main kernel
  bool timing = false;
  static constexpr int kTimingRing = 64;  // launches timed without a host-side wait in between
  hipEvent_t ev0[kTimingRing] = {}, ev1[kTimingRing] = {}, evm[kTimingRing] = {};  // evm: after the first kernel of a split pass
  bool has_mid[kTimingRing] = {};
  double acc_ms = 0.0;
  double acc_first_ms = 0.0;  // first (dominant) kernel of split passes only
  long long first_launches = 0;
  int64_t launches = 0;
  int pending = 0;  // event pairs recorded and not yet read back
  // streamed split pass (psis_loo_impl, pla_kernels.hip launch_wave): two internal streams forked from / joined to the
  // caller's stream -- the wave kernel on one, the fit kernel beside it on the other -- and the flags between them
  hipStream_t pipe_first = nullptr, pipe_second = nullptr;
  hipEvent_t pipe_fork = nullptr, pipe_join1 = nullptr, pipe_join2 = nullptr;
  void* d_sync = nullptr;
  size_t d_sync_bytes = 0;
  // timing of the first kernel alone (bench roofline: average duration of one launch of the dominant kernel)
  static constexpr int kPipeTimed = 64;
  hipEvent_t pipe_t0[kPipeTimed] = {}, pipe_t1[kPipeTimed] = {};
  int pipe_timed = 0;
  std::string last_kernels;  // pla_engine_last_kernels
  // ordering of the calls ACROSS streams: every call that enqueues work on the workspace records `order_event` behind it, and a
  // call on another stream makes that stream wait for it first (EngineCall)
  hipEvent_t order_event = nullptr;
  hipStream_t order_stream = nullptr;
  bool order_valid = false;
};

namespace {

thread_local bool g_frozen = false;  // set from the engine at the start of every entry point (EngineCall)

int grow(void** p, size_t* have, size_t want) {
  if (*have >= want) return PLA_OK;
  if (g_frozen)
    return fail(PLA_ERR_FROZEN, "the engine workspace is frozen (pla_engine_set_frozen) and this call needs %zu more bytes of it",
                want - *have);
  // hipFree waits for the device: launches already enqueued on any stream have finished with the old buffer
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) return fail(PLA_ERR_NOMEM, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
  *have = want;
  return PLA_OK;
}

int check_common(pla_engine* eng, const void* in, int dtype, int64_t n_obs, int64_t n_draws,
                 int64_t stride_obs, int64_t stride_draw, int method, int64_t tail_count, int mem_space) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (dtype != PLA_F64 && dtype != PLA_F32) return fail(PLA_ERR_ARG, "dtype must be PLA_F64 or PLA_F32");
  if (mem_space != PLA_HOST && mem_space != PLA_DEVICE) return fail(PLA_ERR_ARG, "bad mem_space");
  if (method != PLA_PSIS && method != PLA_SIS && method != PLA_TIS) return fail(PLA_ERR_ARG, "bad method");
  if (n_obs < 0) return fail(PLA_ERR_ARG, "n_obs < 0");
  if (n_obs > 0 && !in) return fail(PLA_ERR_ARG, "input pointer is NULL");
  if (n_draws < 1 || n_draws > (int64_t)1 << 30) return fail(PLA_ERR_ARG, "n_draws out of range");
  if (stride_draw == 0 || stride_obs < 0 || stride_draw < 0) return fail(PLA_ERR_ARG, "bad strides");
  if (method == PLA_PSIS) {
    // x_sorted[-M-1] must exist (numpy raises IndexError otherwise: psis.py:136)
    if (tail_count < 1 || tail_count + 1 > n_draws)
      return fail(PLA_ERR_ARG, "tail_count %lld needs at least %lld draws, got %lld", (long long)tail_count,
                  (long long)tail_count + 1, (long long)n_draws);
    if (tail_count > pla::max_tail_count())
      return fail(PLA_ERR_UNSUPPORTED, "tail_count %lld exceeds the LDS tail capacity %d", (long long)tail_count,
                  pla::max_tail_count());
  }
  return PLA_OK;
}

// Row-independent tables of the fast path, computed with the host libm exactly as NumPy does:
//   [0, M)      log1p(-(j + 0.5)/M)              psis.py:153 through log1p of psis.py:219/221
//   [M, M+64)   1 - sqrt(m_est / (j + 0.5))      psis.py:186 for m_est = 30 + isqrt(M)
int ensure_l1_table(pla_engine* e, int64_t M, hipStream_t s, const double** out) {
  for (const auto& t : e->l1_tables)
    if (t.M == M) {
      *out = t.d;
      return PLA_OK;
    }
  if (g_frozen) return fail(PLA_ERR_FROZEN, "the engine is frozen and has no quantile table for tail_count %lld", (long long)M);
  double* d = nullptr;
  hipError_t he = hipMalloc((void**)&d, (size_t)(M + 64) * sizeof(double));
  if (he != hipSuccess) return fail(PLA_ERR_NOMEM, "hipMalloc(table): %s", hipGetErrorString(he));
  std::vector<double> h((size_t)M + 64);
  for (int64_t j = 0; j < M; ++j) h[(size_t)j] = std::log1p(-(((double)j + 0.5) / (double)M));
  int64_t root = (int64_t)std::sqrt((double)M);
  while (root * root > M) --root;
  while ((root + 1) * (root + 1) <= M) ++root;
  const double mest = (double)(30 + root);
  for (int j = 0; j < 64; ++j) h[(size_t)M + j] = 1.0 - std::sqrt(mest / ((double)(j + 1) - 0.5));
  he = hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, s);
  if (he == hipSuccess) he = hipStreamSynchronize(s);  // h goes out of scope
  if (he != hipSuccess) {
    (void)hipFree(d);
    return fail(PLA_ERR_HIP, "table upload: %s", hipGetErrorString(he));
  }
  e->l1_tables.push_back({M, d});
  *out = d;
  return PLA_OK;
}

// every entry point: serialise on the engine and publish its frozen flag to grow().  With a stream: the call enqueues work that
// uses the engine's workspace -- it is ordered behind the previous such call when that one went to ANOTHER stream (one event
// per engine, recorded at the end of every call; nothing inside a stream capture, whose order is the caller's)
struct EngineCall {
  std::lock_guard<std::mutex> lock;
  pla_engine* e;
  hipStream_t s;
  bool ordered;
  explicit EngineCall(pla_engine* e_) : lock(e_->mu), e(e_), s(nullptr), ordered(false) { g_frozen = e->frozen; }
  EngineCall(pla_engine* e_, hipStream_t s_) : lock(e_->mu), e(e_), s(s_), ordered(false) {
    g_frozen = e->frozen;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipSetDevice(e->device) != hipSuccess || !e->order_event) return;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return;
    }
    ordered = true;
    if (e->order_valid && e->order_stream != s) (void)hipStreamWaitEvent(s, e->order_event, 0);
  }
  ~EngineCall() {
    if (ordered && hipEventRecord(e->order_event, s) == hipSuccess) {
      e->order_stream = s;
      e->order_valid = true;
    }
    g_frozen = false;
  }
};

struct TimedLaunch {  // brackets the main kernel with events when timing is on
  pla_engine* e;
  hipStream_t s;
  TimedLaunch(pla_engine* e_, hipStream_t s_) : e(e_), s(s_) {
    if (e->timing) {
      if (e->pending == pla_engine::kTimingRing) flush(e);  // the only case in which the host waits
      e->has_mid[e->pending] = false;  // (a reused ring slot must not inherit the mid event of an earlier split pass)
      (void)hipEventRecord(e->ev0[e->pending], s);
    }
  }
  hipEvent_t mid() const { return e->timing ? e->evm[e->pending] : nullptr; }
  bool* mid_flag() const {
    if (!e->timing) return nullptr;
    e->has_mid[e->pending] = false;
    return &e->has_mid[e->pending];
  }
  ~TimedLaunch() {
    if (e->timing) {
      (void)hipEventRecord(e->ev1[e->pending], s);
      e->pending += 1;
    }
  }
  static void flush(pla_engine* e) {
    for (int i = 0; i < e->pending; ++i) {
      float ms = 0.f;
      if (hipEventSynchronize(e->ev1[i]) == hipSuccess && hipEventElapsedTime(&ms, e->ev0[i], e->ev1[i]) == hipSuccess) {
        e->acc_ms += ms;
        e->launches += 1;
        if (e->has_mid[i] && hipEventElapsedTime(&ms, e->ev0[i], e->evm[i]) == hipSuccess) {
          e->acc_first_ms += ms;
          e->first_launches += 1;
        }
      }
    }
    e->pending = 0;
    for (int i = 0; i < e->pipe_timed; ++i) {
      float ms = 0.f;
      if (hipEventSynchronize(e->pipe_t1[i]) == hipSuccess && hipEventElapsedTime(&ms, e->pipe_t0[i], e->pipe_t1[i]) == hipSuccess) {
        e->acc_first_ms += ms;
        e->first_launches += 1;
      }
    }
    e->pipe_timed = 0;
  }
};

// streams and events of the streamed pass, created on first use
int ensure_pipe(pla_engine* e) {
  if (e->pipe_first) return PLA_OK;
  if (g_frozen) return fail(PLA_ERR_FROZEN, "the engine is frozen and has no internal streams yet");
  hipError_t he = hipStreamCreateWithFlags(&e->pipe_first, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->pipe_second, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e->pipe_fork, hipEventDisableTiming);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e->pipe_join1, hipEventDisableTiming);
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e->pipe_join2, hipEventDisableTiming);
  for (int i = 0; i < pla_engine::kPipeTimed && he == hipSuccess; ++i) {
    he = hipEventCreate(&e->pipe_t0[i]);
    if (he == hipSuccess) he = hipEventCreate(&e->pipe_t1[i]);
  }
  if (he != hipSuccess) return fail(PLA_ERR_HIP, "internal streams / events: %s", hipGetErrorString(he));
  return PLA_OK;
}

}  // namespace

extern "C" {

int pla_abi_version(void) { return PLA_ABI_VERSION; }

const char* pla_last_error(void) { return g_err; }

int pla_device_count(int* count) {
  if (!count) return fail(PLA_ERR_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(PLA_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return PLA_OK;
}

int pla_engine_create(int device, pla_engine** out) {
  if (!out) return fail(PLA_ERR_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PLA_ERR_NODEVICE, "no HIP device available");
  if (device < 0 || device >= n) return fail(PLA_ERR_ARG, "device %d out of range (0..%d)", device, n - 1);
  PLA_HIP(hipSetDevice(device));
  pla_engine* e = new (std::nothrow) pla_engine();
  if (!e) return fail(PLA_ERR_NOMEM, "out of host memory");
  e->device = device;
  hipError_t he = hipMalloc((void**)&e->counters, pla::kCountersTotal * sizeof(unsigned long long));
  if (he == hipSuccess) he = hipMalloc((void**)&e->d_red, (size_t)pla::reduce_workspace_doubles() * sizeof(double));
  if (he == hipSuccess) he = hipMemset(e->counters, 0, pla::kCountersTotal * sizeof(unsigned long long));
  if (he == hipSuccess) he = hipMemset(e->d_red, 0, (size_t)pla::reduce_workspace_doubles() * sizeof(double));  // (the ticket of reduce_fused)
  if (he == hipSuccess) he = hipEventCreateWithFlags(&e->order_event, hipEventDisableTiming);
  for (int i = 0; i < pla_engine::kTimingRing && he == hipSuccess; ++i) {
    he = hipEventCreate(&e->ev0[i]);
    if (he == hipSuccess) he = hipEventCreate(&e->ev1[i]);
    if (he == hipSuccess) he = hipEventCreate(&e->evm[i]);
  }
  if (he != hipSuccess) {
    pla_engine_destroy(e);
    return fail(PLA_ERR_HIP, "engine setup: %s", hipGetErrorString(he));
  }
  *out = e;
  return PLA_OK;
}

int pla_engine_destroy(pla_engine* e) {
  if (!e) return PLA_OK;
  (void)hipSetDevice(e->device);
  if (e->counters) (void)hipFree(e->counters);
  if (e->d_red) (void)hipFree(e->d_red);
  if (e->d_in) (void)hipFree(e->d_in);
  if (e->d_lw) (void)hipFree(e->d_lw);
  if (e->d_pw) (void)hipFree(e->d_pw);
  if (e->d_slow) (void)hipFree(e->d_slow);
  for (auto& t : e->l1_tables) (void)hipFree(t.d);
  if (e->d_ws) (void)hipFree(e->d_ws);
  if (e->d_rows) (void)hipFree(e->d_rows);
  if (e->d_col) (void)hipFree(e->d_col);
  if (e->d_probs) (void)hipFree(e->d_probs);
  if (e->d_slab) (void)hipFree(e->d_slab);
  for (int i = 0; i < pla_engine::kTimingRing; ++i) {
    if (e->ev0[i]) (void)hipEventDestroy(e->ev0[i]);
    if (e->ev1[i]) (void)hipEventDestroy(e->ev1[i]);
    if (e->evm[i]) (void)hipEventDestroy(e->evm[i]);
  }
  if (e->order_event) (void)hipEventDestroy(e->order_event);
  if (e->pipe_first) (void)hipStreamDestroy(e->pipe_first);
  if (e->pipe_second) (void)hipStreamDestroy(e->pipe_second);
  if (e->d_sync) (void)hipFree(e->d_sync);
  for (hipEvent_t ev : {e->pipe_fork, e->pipe_join1, e->pipe_join2})
    if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < pla_engine::kPipeTimed; ++i) {
    if (e->pipe_t0[i]) (void)hipEventDestroy(e->pipe_t0[i]);
    if (e->pipe_t1[i]) (void)hipEventDestroy(e->pipe_t1[i]);
  }
  delete e;
  return PLA_OK;
}

int pla_tail_count(int64_t n_draws, double reff, int64_t* tail_count) {
  if (!tail_count) return fail(PLA_ERR_ARG, "tail_count is NULL");
  if (n_draws < 1 || !(reff > 0.0)) return fail(PLA_ERR_ARG, "need n_draws >= 1 and reff > 0");
  const double a = (double)n_draws / 5.0;
  const double b = 3.0 * std::sqrt((double)n_draws / reff);  // base.py:139-141
  *tail_count = (int64_t)std::ceil(a < b ? a : b);
  return PLA_OK;
}

int pla_engine_set_frozen(pla_engine* e, int frozen) {
  if (!e) return fail(PLA_ERR_ARG, "engine is NULL");
  EngineCall call(e);
  e->frozen = frozen != 0;
  return PLA_OK;
}

int pla_engine_set_timing(pla_engine* e, int enable) {
  if (!e) return fail(PLA_ERR_ARG, "engine is NULL");
  EngineCall call(e);
  TimedLaunch::flush(e);
  e->timing = enable != 0;
  return PLA_OK;
}

int pla_engine_kernel_ms(pla_engine* e, double* total_ms, int64_t* launches) {
  if (!e) return fail(PLA_ERR_ARG, "engine is NULL");
  EngineCall call(e);
  TimedLaunch::flush(e);
  if (total_ms) *total_ms = e->acc_ms;
  if (launches) *launches = e->launches;
  e->acc_ms = 0.0;
  e->launches = 0;
  return PLA_OK;
}

int pla_engine_first_kernel_ms(pla_engine* e, double* total_ms, int64_t* launches) {
  if (!e) return fail(PLA_ERR_ARG, "engine is NULL");
  EngineCall call(e);
  TimedLaunch::flush(e);
  if (total_ms) *total_ms = e->acc_first_ms;
  if (launches) *launches = e->first_launches;
  e->acc_first_ms = 0.0;
  e->first_launches = 0;
  return PLA_OK;
}

int pla_engine_last_kernels(pla_engine* eng, char* buf, int cap) {
  if (!eng || !buf || cap < 1) return fail(PLA_ERR_ARG, "engine / buffer is NULL");
  EngineCall call(eng);
  snprintf(buf, (size_t)cap, "%s", eng->last_kernels.c_str());
  return PLA_OK;
}

int pla_env_overrides(char* buf, int cap) {
  if (!buf || cap < 1) return fail(PLA_ERR_ARG, "buffer is NULL");
  static const char* const kAlways[] = {"PLA_PIPE", "PLA_STREAM_PATIENCE_US", "PLA_FORCE_PATH", "PLA_INGEST_TRANSPOSE", "PLA_INGEST_BLOCK_MB"};
  static const char* const kExperiment[] = {"PLA_DEBUG_SKIP", "PLA_FUSED", "PLA_SKIP_FIT", "PLA_NO_THRESHOLD_CHECK", "PLA_NO_RETRY", "PLA_NO_TILE",
                                            "PLA_TILE_GRID", "PLA_FIT_GRID", "PLA_FIT_HELPERS", "PLA_WAVE_PRIO", "PLA_COL_BLOCK",
                                            "PLA_PRINT_REASONS", "PLA_PRINT_CLOCK"};
  std::string out = pla::kExperiment ? "EXPERIMENT-BUILD" : "";
  const auto add = [&](const char* name) {
    const char* v = getenv(name);
    if (!v) return;
    if (!out.empty()) out += " ";
    out += name;
    out += "=";
    out += v;
  };
  for (const char* n : kAlways) add(n);
  if (pla::kExperiment)
    for (const char* n : kExperiment) add(n);
  snprintf(buf, (size_t)cap, "%s", out.c_str());
  return PLA_OK;
}

int pla_engine_stream_stats(pla_engine* eng, int64_t* gave_up) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  EngineCall call(eng);
  PLA_HIP(hipSetDevice(eng->device));
  PLA_HIP(hipDeviceSynchronize());
  unsigned long long h = 0;
  PLA_HIP(hipMemcpy(&h, eng->counters + pla::kCounterGaveUp, sizeof(h), hipMemcpyDeviceToHost));
  PLA_HIP(hipMemset(eng->counters + pla::kCounterGaveUp, 0, sizeof(h)));
  if (gave_up) *gave_up = (int64_t)h;
  return PLA_OK;
}

int pla_aggregate_pack(pla_engine* eng, const double* agg, int rank, int world, double* table, void* stream) {
  if (!eng || !agg || !table) return fail(PLA_ERR_ARG, "engine / agg / table is NULL");
  if (world < 1 || rank < 0 || rank >= world) return fail(PLA_ERR_ARG, "need 0 <= rank < world");
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  PLA_HIP(pla::launch_aggregate_pack(agg, rank, world, table, (hipStream_t)stream));
  return PLA_OK;
}

int pla_aggregate_merge(pla_engine* eng, const double* table, int world, double* out, void* stream) {
  if (!eng || !table || !out) return fail(PLA_ERR_ARG, "engine / table / out is NULL");
  if (world < 1) return fail(PLA_ERR_ARG, "world < 1");
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  PLA_HIP(pla::launch_aggregate_merge(table, world, out, (hipStream_t)stream));
  return PLA_OK;
}

int pla_reduce_pointwise(pla_engine* eng, const double* diag, const double* loo_i, const double* lppd_i,
                         int64_t n_obs, double good_k, int mem_space, void* stream, double* agg) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (!agg) return fail(PLA_ERR_ARG, "agg is NULL");
  if (n_obs < 0) return fail(PLA_ERR_ARG, "n_obs < 0");
  if (mem_space != PLA_HOST && mem_space != PLA_DEVICE) return fail(PLA_ERR_ARG, "bad mem_space");
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  if (mem_space == PLA_DEVICE) {
    pla::ReduceParams rp{diag, loo_i, lppd_i, n_obs, good_k, agg, nullptr};
    PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
    return PLA_OK;
  }
  const size_t need = (size_t)(3 * n_obs + PLA_AGG_COUNT);
  size_t have_b = eng->d_pw_elems * sizeof(double);
  int rc = grow((void**)&eng->d_pw, &have_b, need * sizeof(double));
  eng->d_pw_elems = have_b / sizeof(double);
  if (rc) return rc;
  double* d = eng->d_pw;
  double *dd = nullptr, *dl = nullptr, *dp = nullptr;
  if (diag) { dd = d; PLA_HIP(hipMemcpyAsync(dd, diag, n_obs * sizeof(double), hipMemcpyHostToDevice, s)); }
  if (loo_i) { dl = d + n_obs; PLA_HIP(hipMemcpyAsync(dl, loo_i, n_obs * sizeof(double), hipMemcpyHostToDevice, s)); }
  if (lppd_i) { dp = d + 2 * n_obs; PLA_HIP(hipMemcpyAsync(dp, lppd_i, n_obs * sizeof(double), hipMemcpyHostToDevice, s)); }
  double* dagg = d + 3 * n_obs;
  pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, dagg, nullptr};
  PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
  PLA_HIP(hipMemcpyAsync(agg, dagg, PLA_AGG_COUNT * sizeof(double), hipMemcpyDeviceToHost, s));
  PLA_HIP(hipStreamSynchronize(s));
  return PLA_OK;
}

// Device matrices with the observations fastest (ArviZ's native layout behind pyloo's stacked view, loo.py:189) are brought
// to draws-fastest row blocks by a tiled transpose kernel and then take the same kernels as everything else.
static bool obs_fastest_device(int mem_space, const int64_t* row_index, int64_t n_src, int64_t n_draws, int64_t stride_obs,
                               int64_t stride_draw) {
  return mem_space == PLA_DEVICE && !row_index && n_src > 1 && n_draws > 1 && stride_obs == 1 && stride_draw >= n_src;
}
// rows per staging block: 1 GiB blocks from the host (pinned-copy granularity), 4 GiB for the device transpose
static int64_t staged_chunk_rows(int mem_space, bool ingest, int64_t n_obs, int64_t n_draws, size_t esz) {
  static const size_t ingest_bytes = [] {  // PLA_INGEST_BLOCK_MB: block size of the transposing ingestion (tuning knob)
    const char* e = getenv("PLA_INGEST_BLOCK_MB");
    const long mb = e ? atol(e) : 0;
    return mb > 0 ? (size_t)mb << 20 : (size_t)1 << 32;
  }();
  const size_t bytes = (mem_space == PLA_DEVICE && ingest) ? ingest_bytes : (size_t)1 << 30;
  int64_t r = (int64_t)(bytes / ((size_t)n_draws * esz));
  if (r < 1) r = 1;
  return r < n_obs ? r : n_obs;
}

// row selection shared by pla_psis_loo_rows / pla_waic_rows: the index list lives where the matrix lives
static int check_rows(const int64_t* row_index, int64_t n_rows, int64_t n_src, int mem_space) {
  if (n_rows < 0) return fail(PLA_ERR_ARG, "n_rows < 0");
  if (n_rows > 0 && !row_index) return fail(PLA_ERR_ARG, "row_index is NULL");
  if (n_rows > 0 && n_src <= 0) return fail(PLA_ERR_ARG, "row selection from an empty matrix");
  if (mem_space == PLA_HOST)
    for (int64_t i = 0; i < n_rows; ++i)
      if (row_index[i] < 0 || row_index[i] >= n_src) return fail(PLA_ERR_ARG, "row_index out of range");
  return PLA_OK;
}

// device path: a clamped copy of the caller's index list in engine memory (the row kernels trust it)
static int device_rows(pla_engine* eng, const int64_t* row_index, int64_t n_rows, int64_t n_src, hipStream_t s,
                       const int64_t** out) {
  int rc = grow(&eng->d_rows, &eng->d_rows_bytes, (size_t)(n_rows > 0 ? n_rows : 1) * sizeof(int64_t));
  if (rc) return rc;
  PLA_HIP(pla::launch_clamp_rows(row_index, n_rows, n_src, (int64_t*)eng->d_rows, s));
  *out = (const int64_t*)eng->d_rows;
  return PLA_OK;
}

// host path: pack rows [r0, r0 + nr) of the call (selected rows when row_index is set) into the staging buffer
static int stage_rows(pla_engine* eng, const void* ll, const int64_t* row_index, int64_t r0, int64_t nr, int64_t stride_obs,
                      size_t esz, size_t row_bytes, hipStream_t s, int64_t stride_draw = 1, int dtype = PLA_F64,
                      int64_t rows_per_chunk = 0) {
  if (stride_draw != 1) {
    // observations fastest on the host (a (chain, draw, *obs) array viewed as (obs, sample), loo.py:189): the block of
    // observations goes up as an (n_draws, nr) slab -- one pitched copy, no transpose on the host -- and is transposed
    // on the device
    const int64_t n_draws = (int64_t)(row_bytes / esz);
    int rc = grow(&eng->d_slab, &eng->d_slab_bytes, (size_t)rows_per_chunk * row_bytes);
    if (rc) return rc;
    PLA_HIP(hipMemcpy2DAsync(eng->d_slab, (size_t)nr * esz, (const char*)ll + (size_t)r0 * esz, (size_t)stride_draw * esz,
                             (size_t)nr * esz, (size_t)n_draws, hipMemcpyHostToDevice, s));
    PLA_HIP(pla::launch_transpose_rows(eng->d_slab, dtype, nr, 0, nr, (int)n_draws, eng->d_in, s));
    return PLA_OK;
  }
  if (!row_index) {
    const char* src = (const char*)ll + (size_t)r0 * stride_obs * esz;
    PLA_HIP(hipMemcpy2DAsync(eng->d_in, row_bytes, src, (size_t)stride_obs * esz, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
  } else {
    for (int64_t i = 0; i < nr; ++i)
      PLA_HIP(hipMemcpyAsync((char*)eng->d_in + (size_t)i * row_bytes, (const char*)ll + (size_t)row_index[r0 + i] * stride_obs * esz,
                             row_bytes, hipMemcpyHostToDevice, s));
  }
  return PLA_OK;
}

static int psis_loo_impl(pla_engine* eng, const void* ll, int dtype, int64_t n_src, const int64_t* row_index, int64_t n_obs,
                         int64_t n_draws, int64_t stride_obs, int64_t stride_draw, int method, int64_t tail_count,
                         double scale_value, double good_k, int mem_space, void* stream, double* diag, double* loo_i,
                         double* lppd_i, double* agg) {
  // (n_obs = observations of this call: the selected rows when row_index is set, else all n_src rows)
  int rc = check_common(eng, ll, dtype, n_src, n_draws, stride_obs, stride_draw, method, tail_count, mem_space);
  if (rc) return rc;
  if (row_index || n_obs != n_src) {
    rc = check_rows(row_index, n_obs, n_src, mem_space);
    if (rc) return rc;
  }
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == PLA_F64 ? 8 : 4;
  const bool ingest = obs_fastest_device(mem_space, row_index, n_src, n_draws, stride_obs, stride_draw);
  // observations-fastest PSIS-LOO: the lane-per-observation kernels read the matrix as it lies (pla_col.h); PLA_INGEST_TRANSPOSE=1
  // keeps round 1's transposing ingestion (A/B runs), which also serves the shapes the column kernels do not take
  static const int64_t kColBlock = [] {  // observations per launch: 8 KB of candidate lists each (PLA_COL_BLOCK: A/B runs, experiment builds only)
    const char* e = pla::exp_str("PLA_COL_BLOCK");
    const int64_t v = e ? atoll(e) : 0;
    // (at most 262 144: the lists of one launch are addressed with 32-bit byte offsets, 8 KB per observation)
    return v >= 256 ? (v <= 262144 ? v : (int64_t)262144) : (int64_t)262144;
  }();
  int col_kq = 0;
  static const bool force_transpose = getenv("PLA_INGEST_TRANSPOSE") && atoi(getenv("PLA_INGEST_TRANSPOSE")) != 0;
  constexpr int64_t kDevBlock = (int64_t)1 << 20;  // rows per launch of a device-resident matrix (bounds the hand-over buffer)
  // (the column sweep addresses the draws of a batch with 32-bit byte offsets from the batch's first draw -- sixteen draws at
  // most -- plus the lane's observation inside the block: matrices whose draws lie further apart than that allows take the
  // transposing path instead of reading zeros past a descriptor's range)
  const bool col_offsets_fit = (double)stride_draw * (double)esz * 16.0 + (double)kColBlock * (double)esz < 2147483648.0;
  const bool use_col = ingest && method == PLA_PSIS && !force_transpose && col_offsets_fit &&
                       pla::col_supported((int)n_draws, (int)tail_count, &col_kq);
  // ... a workgroup per 16 observations (pla_tile.h), streamed (the fit kernel beside it, PLA_PIPE=0: back to back) where the
  // shorter lists of the streamed pass still leave room around the expected candidate count
  const char* pipe_env0 = getenv("PLA_PIPE");
  int tile_ks = 0;
  bool tile_stream = !(pipe_env0 && atoi(pipe_env0) == 0) && use_col && mem_space == PLA_DEVICE &&
                     pla::tile_supported(dtype, (int)n_draws, (int)tail_count, stride_draw, true, &tile_ks);
  const bool use_tile = tile_stream || (use_col && pla::tile_supported(dtype, (int)n_draws, (int)tail_count, stride_draw, false, &tile_ks));
  // Streamed split pass (device-resident, draws-fastest matrices): the fit kernel runs BESIDE the wave kernel, in the registers
  // and LDS that kernel leaves free on a CU, and takes the chunks of observations as they are finished (pla_kernels.hip,
  // launch_wave).  PLA_PIPE=0: the two kernels back to back on the caller's stream (round 2's arrangement; A/B runs).
  const char* pipe_env = getenv("PLA_PIPE");  // (read per call: tests compare the two arrangements in one process)
  bool pipeline = !(pipe_env && atoi(pipe_env) == 0) && mem_space == PLA_DEVICE && !ingest && method == PLA_PSIS;

  pla::RowsParams p{};
  p.n_obs = n_obs;
  p.n_draws = (int)n_draws;
  p.method = method;
  p.tail_count = method == PLA_PSIS ? (int)tail_count : 0;
  p.tail_cap = pow2_at_least(p.tail_count);
  p.scale_value = scale_value;
  p.counters = eng->counters;
  if (n_obs > 0) {
    rc = grow(&eng->d_slow, &eng->d_slow_bytes, (size_t)n_obs * sizeof(unsigned));
    if (rc) return rc;
    p.slow_list = (unsigned*)eng->d_slow;
    if (method == PLA_PSIS) {
      rc = ensure_l1_table(eng, tail_count, s, &p.l1_table);
      if (rc) return rc;
      // hand-over buffers of the split pass (one-chunk wave kernel -> fit kernel, pla_fit.h): sized for the
      // rows one launch processes (all of them on the device path, one staging chunk on the host path)
      if ((tail_count <= 250 && n_draws >= 256 && n_draws <= 4096) || (tail_count <= 448 && n_draws >= 256)) {  // (one-chunk / chunked wave kernels)
        const int64_t col_rows = use_tile ? kDevBlock : kColBlock;
        const int64_t chunk_rows = use_col ? (n_obs < col_rows ? n_obs : col_rows) : staged_chunk_rows(mem_space, ingest, n_obs, n_draws, esz);
        // (device-resident matrices run in blocks of kDevBlock rows, so the buffer is bounded: 1.7 GB at M = 190 however
        // many observations there are)
        int64_t rows = (mem_space == PLA_DEVICE && !ingest) ? (n_obs < kDevBlock ? n_obs : kDevBlock) : chunk_rows;
        const int stride = (int)((tail_count + 63) & ~(int64_t)63);  // 16 lanes x 4 values per quad
        if (pipeline) {  // does the launcher stream these rows at all?
          pla::RowsParams q = p;
          q.in = ll;
          q.stride_obs = stride_obs;
          q.stride_draw = stride_draw;
          q.n_obs = rows;
          q.ws_y = q.ws_s = (double*)eng->counters;  // (any non-null pointer: only looked at, never dereferenced)
          q.ws_stride = stride;
          q.ws_sstride = 16;
          pipeline = pla::rows_stream_planned(q, dtype);
        }
        if (tile_stream && !(stride <= 256)) tile_stream = false;
        if (pipeline || tile_stream) {
          rc = ensure_pipe(eng);
          if (rc) return rc;
          rc = grow(&eng->d_sync, &eng->d_sync_bytes, pla::stream_sync_bytes(rows));
          if (rc) return rc;
        }
        const int sstride = (pipeline || tile_stream) ? 16 : 8;  // (streamed: one whole 128-byte line of scalars per observation)
        rc = grow(&eng->d_ws, &eng->d_ws_bytes, (size_t)rows * (size_t)(stride + sstride) * sizeof(double));
        if (rc == PLA_ERR_NOMEM && !use_col) {
          rc = 0;  // no room for the hand-over: the fused kernels need none (the split pass is the faster, not the only, path)
        } else {
          if (rc) return rc;
          p.ws_y = (double*)eng->d_ws;
          p.ws_s = (double*)eng->d_ws + (size_t)rows * stride;
          p.ws_stride = stride;
          p.ws_sstride = sstride;
        }
      }
    }
  }
  if (!p.ws_y) pipeline = tile_stream = false;
  if (!use_tile) tile_stream = false;
  if (use_tile && !tile_stream) (void)pla::tile_supported(dtype, (int)n_draws, (int)tail_count, stride_draw, false, &tile_ks);  // (the longer lists' threshold)
  // [1]: rows left to the general kernel (a streamed pass zeroes the counters together with its flags: one command less)
  if (!pipeline && !tile_stream) PLA_HIP(hipMemsetAsync(eng->counters, 0, pla::kCountersPerCall * sizeof(unsigned long long), s));

  if (mem_space == PLA_DEVICE) {
    // agg needs the pointwise loo_i: use the caller's vectors, or the engine scratch
    double *dd = diag, *dl = loo_i, *dp = lppd_i;
    if (agg && (!dd || !dl || !dp)) {
      size_t have_b = eng->d_pw_elems * sizeof(double);
      rc = grow((void**)&eng->d_pw, &have_b, (size_t)(3 * n_obs + PLA_AGG_COUNT) * sizeof(double));
      eng->d_pw_elems = have_b / sizeof(double);
      if (rc) return rc;
      if (!dd) dd = eng->d_pw;
      if (!dl) dl = eng->d_pw + n_obs;
      if (!dp) dp = eng->d_pw + 2 * n_obs;
    }
    if (use_tile && p.ws_y) {
      // observations-fastest input, read in place: a workgroup per 16 observations, candidate lists in LDS (pla_tile.h)
      p.stride_obs = 1;
      p.stride_draw = stride_draw;
      for (int64_t r0 = 0; r0 < n_obs; r0 += kDevBlock) {
        const int64_t nr = (n_obs - r0 < kDevBlock) ? (n_obs - r0) : kDevBlock;
        TimedLaunch t(eng, s);
        p.in = (const char*)ll + (size_t)r0 * esz;
        p.n_obs = nr;
        p.diag = dd ? dd + r0 : nullptr;
        p.loo_i = dl ? dl + r0 : nullptr;
        p.lppd_i = dp ? dp + r0 : nullptr;
        if (tile_stream) {
          pla::PipeStreams ps{eng->pipe_first, eng->pipe_second, eng->pipe_fork, eng->pipe_join1, eng->pipe_join2, (unsigned*)eng->d_sync,
                              nullptr, nullptr, r0 == 0};
          if (eng->timing && eng->pipe_timed < pla_engine::kPipeTimed) {
            ps.before_first = eng->pipe_t0[eng->pipe_timed];
            ps.after_first = eng->pipe_t1[eng->pipe_timed];
            eng->pipe_timed += 1;
          }
          PLA_HIP(pla::launch_tile(p, dtype, tile_ks, s, &ps));
          eng->last_kernels = "tile_loo_kernel<SYNC> (a workgroup per 16 observations, matrix read in place) with fit_rows_stream_kernel beside it "
                              "on a second stream + fit_rows_kernel (leftovers) + slow_rows_kernel";
        } else {
          PLA_HIP(pla::launch_tile(p, dtype, tile_ks, s));
          eng->last_kernels = "tile_loo_kernel (a workgroup per 16 observations, matrix read in place) + fit_rows_kernel + slow_rows_kernel";
        }
      }
      if (agg) {
        pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, agg, eng->counters + 1};
        PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
      }
      return PLA_OK;
    }
    if (use_col && p.ws_y) {
      // observations-fastest input, read in place: one lane per observation (pla_col.h), block by block of observations
      rc = grow(&eng->d_col, &eng->d_col_bytes, pla::col_workspace_bytes(n_obs < kColBlock ? n_obs : kColBlock));
      if (rc) return rc;
      p.stride_obs = 1;
      p.stride_draw = stride_draw;
      for (int64_t r0 = 0; r0 < n_obs; r0 += kColBlock) {
        const int64_t nr = (n_obs - r0 < kColBlock) ? (n_obs - r0) : kColBlock;
        TimedLaunch t(eng, s);
        p.in = (const char*)ll + (size_t)r0 * esz;
        p.n_obs = nr;
        p.diag = dd ? dd + r0 : nullptr;
        p.loo_i = dl ? dl + r0 : nullptr;
        p.lppd_i = dp ? dp + r0 : nullptr;
        PLA_HIP(pla::launch_col(p, dtype, col_kq, eng->d_col, s));
        eng->last_kernels = "col_sweep_kernel (one lane per observation, matrix read in place) + col_select_kernel + fit_rows_kernel + slow_rows_kernel";
      }
      if (agg) {
        pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, agg, eng->counters + 1};
        PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
      }
      return PLA_OK;
    }
    if (ingest) {
      // observations-fastest input: transpose a block of rows into the staging buffer, run the pass on it, next block
      // (all on the caller's stream: the buffer is reused in stream order, nothing synchronises)
      const int64_t rows_per_chunk = staged_chunk_rows(mem_space, true, n_obs, n_draws, esz);
      rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * (size_t)n_draws * esz);
      if (rc) return rc;
      p.in = eng->d_in;
      p.stride_obs = n_draws;
      p.stride_draw = 1;
      for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
        const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
        TimedLaunch t(eng, s);
        PLA_HIP(pla::launch_transpose_rows(ll, dtype, stride_draw, r0, nr, (int)n_draws, eng->d_in, s));
        p.n_obs = nr;
        p.diag = dd ? dd + r0 : nullptr;
        p.loo_i = dl ? dl + r0 : nullptr;
        p.lppd_i = dp ? dp + r0 : nullptr;
        PLA_HIP(pla::launch_rows(p, dtype, false, s));
      eng->last_kernels = pla::last_rows_kernels();
      }
      if (agg) {
        pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, agg, eng->counters + 1};
        PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
      }
      return PLA_OK;
    }
    p.in = ll;
    p.stride_obs = stride_obs;
    p.stride_draw = stride_draw;
    if (row_index) {
      rc = device_rows(eng, row_index, n_obs, n_src, s, &p.row_index);
      if (rc) return rc;
    }
    const int64_t* all_rows = p.row_index;
    for (int64_t r0 = 0; r0 < n_obs; r0 += kDevBlock) {
      const int64_t nr = (n_obs - r0 < kDevBlock) ? (n_obs - r0) : kDevBlock;
      p.n_obs = nr;
      if (all_rows) p.row_index = all_rows + r0;
      else p.in = (const char*)ll + (size_t)r0 * (size_t)stride_obs * esz;
      p.diag = dd ? dd + r0 : nullptr;
      p.loo_i = dl ? dl + r0 : nullptr;
      p.lppd_i = dp ? dp + r0 : nullptr;
      TimedLaunch t(eng, s);
      if (pipeline) {
        pla::PipeStreams ps{eng->pipe_first, eng->pipe_second, eng->pipe_fork, eng->pipe_join1, eng->pipe_join2, (unsigned*)eng->d_sync,
                            nullptr, nullptr, r0 == 0};
        if (eng->timing && eng->pipe_timed < pla_engine::kPipeTimed) {
          ps.before_first = eng->pipe_t0[eng->pipe_timed];
          ps.after_first = eng->pipe_t1[eng->pipe_timed];
          eng->pipe_timed += 1;
        }
        PLA_HIP(pla::launch_rows(p, dtype, false, s, nullptr, nullptr, &ps));
      eng->last_kernels = pla::last_rows_kernels();
      } else {
        PLA_HIP(pla::launch_rows(p, dtype, false, s, t.mid(), t.mid_flag()));
      eng->last_kernels = pla::last_rows_kernels();
      }
    }
#if defined(PLA_WAVE_ABLATE) && PLA_WAVE_ABLATE
    if (pla::exp_str("PLA_PRINT_REASONS")) {  // profiling build only: why rows left the fast path
      unsigned long long h[16];
      PLA_HIP(hipStreamSynchronize(s));
      PLA_HIP(hipMemcpy(h, eng->counters, sizeof(h), hipMemcpyDeviceToHost));
      fprintf(stderr, "[pla] slow rows by reason: range %llu, threshold search %llu, t1>=0 %llu, pads %llu, too few candidates %llu, "
                      "too many %llu, other %llu, selection/fit %llu\n", h[8], h[9], h[10], h[11], h[12], h[13], h[14], h[15]);
    }
    if (pla::exp_str("PLA_PRINT_CLOCK")) {  // profiling build only: core clock seen by one wave of the fast kernel
      unsigned long long h[4];
      PLA_HIP(hipStreamSynchronize(s));
      PLA_HIP(hipMemcpy(h, eng->counters, sizeof(h), hipMemcpyDeviceToHost));
      if (h[3]) fprintf(stderr, "[pla] core clock %.1f MHz (%llu core ticks / %llu ticks of 100 MHz)\n",
                        100.0 * (double)h[2] / (double)h[3], h[2], h[3]);
    }
#endif
#if defined(PLA_PHASE_CLOCK)
    {  // diagnostic build: cycles per row of the wave kernel's waves, by phase (tools/phase_clock.sh)
      unsigned long long h[16];
      PLA_HIP(hipStreamSynchronize(s));
      PLA_HIP(hipMemcpy(h, eng->counters, sizeof(h), hipMemcpyDeviceToHost));
      const double rows = h[12] ? (double)h[12] : 1.0;
      fprintf(stderr, "[pla] cycles per row and wave: statistics (incl. wait for the row) %.0f, threshold %.0f, sweep %.0f, selection %.0f; rows %llu\n",
              h[8] / rows, h[9] / rows, h[10] / rows, h[11] / rows, h[12]);
    }
#endif
    if (agg) {
      pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, agg, eng->counters + 1};
      PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
    }
    return PLA_OK;
  }

  // ---- PLA_HOST: stage row blocks through the device --------------------------------------
  {
    size_t have_b = eng->d_pw_elems * sizeof(double);
    rc = grow((void**)&eng->d_pw, &have_b, (size_t)(3 * n_obs + PLA_AGG_COUNT) * sizeof(double));
    eng->d_pw_elems = have_b / sizeof(double);
    if (rc) return rc;
  }
  double* dd = eng->d_pw;
  double* dl = eng->d_pw + n_obs;
  double* dp = eng->d_pw + 2 * n_obs;
  double* dagg = eng->d_pw + 3 * n_obs;
  const size_t row_bytes = (size_t)n_draws * esz;
  int64_t rows_per_chunk = (int64_t)(((size_t)1 << 30) / (row_bytes ? row_bytes : 1));
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  if (rows_per_chunk > n_obs) rows_per_chunk = n_obs;
  if (n_obs > 0) {
    rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * row_bytes);
    if (rc) return rc;
  }
  for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
    const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
    // pack to (nr, S) contiguous on the device; strided sources use a pitched copy
    if (stride_draw == 1 || (stride_obs == 1 && stride_draw >= n_src && !row_index)) {
      rc = stage_rows(eng, ll, row_index, r0, nr, stride_obs, esz, row_bytes, s, stride_draw, dtype, rows_per_chunk);
      if (rc) return rc;
      p.stride_obs = n_draws;
      p.stride_draw = 1;
    } else {
      return fail(PLA_ERR_UNSUPPORTED, "PLA_HOST input needs unit stride along the draws, or along the observations without "
                                       "a row selection");
    }
    p.in = eng->d_in;
    p.n_obs = nr;
    p.diag = dd + r0;
    p.loo_i = dl + r0;
    p.lppd_i = dp + r0;
    {
      TimedLaunch t(eng, s);
      PLA_HIP(pla::launch_rows(p, dtype, false, s));
      eng->last_kernels = pla::last_rows_kernels();
    }
    PLA_HIP(hipStreamSynchronize(s));  // the staging buffer is reused by the next chunk
  }
  if (n_obs > 0) {
    if (diag) PLA_HIP(hipMemcpyAsync(diag, dd, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
    if (loo_i) PLA_HIP(hipMemcpyAsync(loo_i, dl, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
    if (lppd_i) PLA_HIP(hipMemcpyAsync(lppd_i, dp, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (agg) {
    pla::ReduceParams rp{dd, dl, dp, n_obs, good_k, dagg, eng->counters + 1};
    PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
    PLA_HIP(hipMemcpyAsync(agg, dagg, PLA_AGG_COUNT * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  PLA_HIP(hipStreamSynchronize(s));
  return PLA_OK;
}

int pla_psis_loo(pla_engine* eng, const void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t stride_obs,
                 int64_t stride_draw, int method, int64_t tail_count, double scale_value, double good_k,
                 int mem_space, void* stream, double* diag, double* loo_i, double* lppd_i, double* agg) {
  return psis_loo_impl(eng, ll, dtype, n_obs, nullptr, n_obs, n_draws, stride_obs, stride_draw, method, tail_count, scale_value,
                       good_k, mem_space, stream, diag, loo_i, lppd_i, agg);
}

int pla_psis_loo_rows(pla_engine* eng, const void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t stride_obs,
                      int64_t stride_draw, const int64_t* row_index, int64_t n_rows, int method, int64_t tail_count,
                      double scale_value, double good_k, int mem_space, void* stream, double* diag, double* loo_i,
                      double* lppd_i, double* agg) {
  if (n_rows > 0 && !row_index) return fail(PLA_ERR_ARG, "row_index is NULL");
  return psis_loo_impl(eng, ll, dtype, n_obs, row_index, n_rows, n_draws, stride_obs, stride_draw, method, tail_count,
                       scale_value, good_k, mem_space, stream, diag, loo_i, lppd_i, agg);
}

int pla_importance_weights(pla_engine* eng, const void* logw, int dtype, int64_t n_obs, int64_t n_draws,
                           int64_t stride_obs, int64_t stride_draw, int method, int64_t tail_count,
                           int mem_space, void* stream, void* lw_out, double* diag) {
  int rc = check_common(eng, logw, dtype, n_obs, n_draws, stride_obs, stride_draw, method, tail_count, mem_space);
  if (rc) return rc;
  if (n_obs > 0 && !lw_out) return fail(PLA_ERR_ARG, "lw_out is NULL");
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == PLA_F64 ? 8 : 4;

  pla::RowsParams p{};
  p.n_obs = n_obs;
  p.n_draws = (int)n_draws;
  p.method = method;
  p.tail_count = method == PLA_PSIS ? (int)tail_count : 0;
  p.tail_cap = pow2_at_least(p.tail_count);
  p.scale_value = 1.0;
  p.counters = eng->counters;
  PLA_HIP(hipMemsetAsync(eng->counters, 0, pla::kCountersPerCall * sizeof(unsigned long long), s));
  if (n_obs > 0) {  // workspace of the fast path (rows it declines, quantile tables)
    rc = grow(&eng->d_slow, &eng->d_slow_bytes, (size_t)n_obs * sizeof(unsigned));
    if (rc) return rc;
    p.slow_list = (unsigned*)eng->d_slow;
    if (method == PLA_PSIS) {
      rc = ensure_l1_table(eng, tail_count, s, &p.l1_table);
      if (rc) return rc;
    }
  }

  if (mem_space == PLA_DEVICE) {
    if (obs_fastest_device(mem_space, nullptr, n_obs, n_draws, stride_obs, stride_draw)) {
      // observations-fastest log ratios: transposed block by block like the LOO pass; the weights come out as the
      // (n_obs, n_draws) C-contiguous matrix the header promises
      const int64_t rows_per_chunk = staged_chunk_rows(mem_space, true, n_obs, n_draws, esz);
      rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * (size_t)n_draws * esz);
      if (rc) return rc;
      p.in = eng->d_in;
      p.stride_obs = n_draws;
      p.stride_draw = 1;
      TimedLaunch t(eng, s);
      for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
        const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
        PLA_HIP(pla::launch_transpose_rows(logw, dtype, stride_draw, r0, nr, (int)n_draws, eng->d_in, s));
        p.n_obs = nr;
        p.diag = diag ? diag + r0 : nullptr;
        p.lw_out = (char*)lw_out + (size_t)r0 * (size_t)n_draws * esz;
        PLA_HIP(pla::launch_rows(p, dtype, true, s));
      eng->last_kernels = pla::last_rows_kernels();
      }
      return PLA_OK;
    }
    p.in = logw;
    p.stride_obs = stride_obs;
    p.stride_draw = stride_draw;
    p.diag = diag;
    p.lw_out = lw_out;
    // Rows longer than the registers (S > 4096) with tails the fit kernel takes: the split weights pass (selection kernel -> fit
    // kernel -> output kernel, csrc/pla_lwout.h) needs the hand-over buffers -- the tail's x and eight scalars per observation -- for
    // the rows of one launch, so such calls run in blocks of 2^17 observations (0.48 GB at 448-value tails).  No room: the
    // fused weights kernel, which needs none.
    int64_t block = n_obs;
    if (method == PLA_PSIS && n_draws > 4096 && tail_count <= 448 && stride_draw == 1 && n_obs > 0) {
      const int64_t kLwBlock = (int64_t)1 << 17;
      block = n_obs < kLwBlock ? n_obs : kLwBlock;
      const int stride = (int)((tail_count + 63) & ~(int64_t)63);
      rc = grow(&eng->d_ws, &eng->d_ws_bytes, (size_t)block * (size_t)(stride + 8) * sizeof(double));
      if (rc == PLA_ERR_NOMEM) {
        block = n_obs;
      } else {
        if (rc) return rc;
        p.ws_y = (double*)eng->d_ws;
        p.ws_s = p.ws_y + (size_t)block * stride;
        p.ws_stride = stride;
        p.ws_sstride = 8;
        p.lw_split = true;
      }
    }
    TimedLaunch t(eng, s);
    for (int64_t r0 = 0; r0 < n_obs || r0 == 0; r0 += block) {
      const int64_t nr = (n_obs - r0 < block) ? (n_obs - r0) : block;
      p.in = (const char*)logw + (size_t)r0 * (size_t)stride_obs * esz;
      p.n_obs = nr;
      p.diag = diag ? diag + r0 : nullptr;
      p.lw_out = (char*)lw_out + (size_t)r0 * (size_t)n_draws * esz;
      PLA_HIP(pla::launch_rows(p, dtype, true, s));
      eng->last_kernels = pla::last_rows_kernels();
      if (n_obs == 0) break;
    }
    return PLA_OK;
  }

  if (stride_draw != 1) return fail(PLA_ERR_UNSUPPORTED, "PLA_HOST input needs stride_draw == 1 (transpose on the host)");
  const size_t row_bytes = (size_t)n_draws * esz;
  int64_t rows_per_chunk = (int64_t)(((size_t)1 << 30) / (row_bytes ? row_bytes : 1));
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  if (rows_per_chunk > n_obs) rows_per_chunk = n_obs;
  if (n_obs == 0) return PLA_OK;
  rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * row_bytes);
  if (rc) return rc;
  rc = grow(&eng->d_lw, &eng->d_lw_bytes, (size_t)rows_per_chunk * row_bytes);
  if (rc) return rc;
  if (method == PLA_PSIS && n_draws > 4096 && tail_count <= 448) {  // (the split weights pass of long rows, as on the device path)
    const int stride = (int)((tail_count + 63) & ~(int64_t)63);
    rc = grow(&eng->d_ws, &eng->d_ws_bytes, (size_t)rows_per_chunk * (size_t)(stride + 8) * sizeof(double));
    if (rc && rc != PLA_ERR_NOMEM) return rc;
    if (!rc) {
      p.ws_y = (double*)eng->d_ws;
      p.ws_s = p.ws_y + (size_t)rows_per_chunk * stride;
      p.ws_stride = stride;
      p.ws_sstride = 8;
      p.lw_split = true;
    }
  }
  {
    size_t have_b = eng->d_pw_elems * sizeof(double);
    rc = grow((void**)&eng->d_pw, &have_b, (size_t)(3 * n_obs + PLA_AGG_COUNT) * sizeof(double));
    eng->d_pw_elems = have_b / sizeof(double);
    if (rc) return rc;
  }
  for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
    const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
    const char* src = (const char*)logw + (size_t)r0 * stride_obs * esz;
    PLA_HIP(hipMemcpy2DAsync(eng->d_in, row_bytes, src, (size_t)stride_obs * esz, row_bytes, (size_t)nr,
                             hipMemcpyHostToDevice, s));
    p.in = eng->d_in;
    p.stride_obs = n_draws;
    p.stride_draw = 1;
    p.n_obs = nr;
    p.diag = eng->d_pw + r0;
    p.lw_out = eng->d_lw;
    {
      TimedLaunch t(eng, s);
      PLA_HIP(pla::launch_rows(p, dtype, true, s));
      eng->last_kernels = pla::last_rows_kernels();
    }
    PLA_HIP(hipMemcpyAsync((char*)lw_out + (size_t)r0 * row_bytes, eng->d_lw, (size_t)nr * row_bytes,
                           hipMemcpyDeviceToHost, s));
    PLA_HIP(hipStreamSynchronize(s));
  }
  if (diag) PLA_HIP(hipMemcpyAsync(diag, eng->d_pw, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
  PLA_HIP(hipStreamSynchronize(s));
  return PLA_OK;
}

static int waic_impl(pla_engine* eng, const void* ll, int dtype, int64_t n_src, const int64_t* row_index, int64_t n_obs,
                     int64_t n_draws, int64_t stride_obs, int64_t stride_draw, double scale_value, int mem_space, void* stream,
                     double* lppd_i, double* var_i, double* waic_i, double* agg) {
  int rc = check_common(eng, ll, dtype, n_src, n_draws, stride_obs, stride_draw, PLA_SIS, 0, mem_space);
  if (rc) return rc;
  if (row_index || n_obs != n_src) {
    rc = check_rows(row_index, n_obs, n_src, mem_space);
    if (rc) return rc;
  }
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  const size_t esz = dtype == PLA_F64 ? 8 : 4;
  PLA_HIP(hipMemsetAsync(eng->counters, 0, pla::kCountersPerCall * sizeof(unsigned long long), s));  // [1]: replaced entries

  if (mem_space == PLA_DEVICE) {
    double *dl = lppd_i, *dv = var_i, *dw = waic_i;
    if (agg && (!dv || !dw)) {  // the aggregates need var_i and waic_i: engine scratch when not asked for
      size_t have_b = eng->d_pw_elems * sizeof(double);
      rc = grow((void**)&eng->d_pw, &have_b, (size_t)(3 * n_obs + PLA_AGG_COUNT) * sizeof(double));
      eng->d_pw_elems = have_b / sizeof(double);
      if (rc) return rc;
      if (!dv) dv = eng->d_pw;
      if (!dw) dw = eng->d_pw + n_obs;
    }
    static const bool force_transpose = getenv("PLA_INGEST_TRANSPOSE") && atoi(getenv("PLA_INGEST_TRANSPOSE")) != 0;
    if (obs_fastest_device(mem_space, row_index, n_src, n_draws, stride_obs, stride_draw) && !force_transpose) {
      // observations-fastest matrix, read in place: one lane per observation (pla_waic.h)
      TimedLaunch t(eng, s);
      PLA_HIP(pla::launch_waic_col(ll, dtype, n_obs, (int)n_draws, stride_draw, scale_value, dl, dv, dw, eng->counters + 1, s));
    } else if (obs_fastest_device(mem_space, row_index, n_src, n_draws, stride_obs, stride_draw)) {
      const int64_t rows_per_chunk = staged_chunk_rows(mem_space, true, n_obs, n_draws, esz);
      rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * (size_t)n_draws * esz);
      if (rc) return rc;
      for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
        const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
        TimedLaunch t(eng, s);
        PLA_HIP(pla::launch_transpose_rows(ll, dtype, stride_draw, r0, nr, (int)n_draws, eng->d_in, s));
        PLA_HIP(pla::launch_waic(eng->d_in, nullptr, dtype, nr, (int)n_draws, n_draws, 1, scale_value, dl ? dl + r0 : nullptr,
                                 dv ? dv + r0 : nullptr, dw ? dw + r0 : nullptr, eng->counters + 1, s));
      }
    } else {
      TimedLaunch t(eng, s);
      const int64_t* rows_d = nullptr;
      if (row_index) {
        rc = device_rows(eng, row_index, n_obs, n_src, s, &rows_d);
        if (rc) return rc;
      }
      PLA_HIP(pla::launch_waic(ll, rows_d, dtype, n_obs, (int)n_draws, stride_obs, stride_draw, scale_value, dl, dv, dw,
                               eng->counters + 1, s));
    }
    if (agg) {
      pla::ReduceParams rp{dv, dw, dv, n_obs, 0.4, agg, eng->counters + 1};  // waic.py:147 threshold
      PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
    }
    return PLA_OK;
  }

  if (stride_draw != 1 && !(stride_obs == 1 && stride_draw >= n_src && !row_index))
    return fail(PLA_ERR_UNSUPPORTED, "PLA_HOST input needs unit stride along the draws, or along the observations without a "
                                     "row selection");
  {
    size_t have_b = eng->d_pw_elems * sizeof(double);
    rc = grow((void**)&eng->d_pw, &have_b, (size_t)(3 * n_obs + PLA_AGG_COUNT) * sizeof(double));
    eng->d_pw_elems = have_b / sizeof(double);
    if (rc) return rc;
  }
  double* dl = eng->d_pw;
  double* dv = eng->d_pw + n_obs;
  double* dw = eng->d_pw + 2 * n_obs;
  double* dagg = eng->d_pw + 3 * n_obs;
  const size_t row_bytes = (size_t)n_draws * esz;
  int64_t rows_per_chunk = (int64_t)(((size_t)1 << 30) / (row_bytes ? row_bytes : 1));
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  if (rows_per_chunk > n_obs) rows_per_chunk = n_obs;
  if (n_obs > 0) {
    rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * row_bytes);
    if (rc) return rc;
  }
  for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
    const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
    rc = stage_rows(eng, ll, row_index, r0, nr, stride_obs, esz, row_bytes, s, stride_draw, dtype, rows_per_chunk);
    if (rc) return rc;
    {
      TimedLaunch t(eng, s);
      PLA_HIP(pla::launch_waic(eng->d_in, nullptr, dtype, nr, (int)n_draws, n_draws, 1, scale_value, dl + r0, dv + r0, dw + r0,
                               eng->counters + 1, s));
    }
    PLA_HIP(hipStreamSynchronize(s));  // the staging buffer is reused by the next chunk
  }
  if (n_obs > 0) {
    if (lppd_i) PLA_HIP(hipMemcpyAsync(lppd_i, dl, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
    if (var_i) PLA_HIP(hipMemcpyAsync(var_i, dv, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
    if (waic_i) PLA_HIP(hipMemcpyAsync(waic_i, dw, n_obs * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (agg) {
    pla::ReduceParams rp{dv, dw, dv, n_obs, 0.4, dagg, eng->counters + 1};
    PLA_HIP(pla::launch_reduce(rp, eng->d_red, s));
    PLA_HIP(hipMemcpyAsync(agg, dagg, PLA_AGG_COUNT * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  PLA_HIP(hipStreamSynchronize(s));
  return PLA_OK;
}

int pla_waic(pla_engine* eng, const void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t stride_obs,
             int64_t stride_draw, double scale_value, int mem_space, void* stream, double* lppd_i, double* var_i,
             double* waic_i, double* agg) {
  return waic_impl(eng, ll, dtype, n_obs, nullptr, n_obs, n_draws, stride_obs, stride_draw, scale_value, mem_space, stream,
                   lppd_i, var_i, waic_i, agg);
}

int pla_waic_rows(pla_engine* eng, const void* ll, int dtype, int64_t n_obs, int64_t n_draws, int64_t stride_obs,
                  int64_t stride_draw, const int64_t* row_index, int64_t n_rows, double scale_value, int mem_space,
                  void* stream, double* lppd_i, double* var_i, double* waic_i, double* agg) {
  if (n_rows > 0 && !row_index) return fail(PLA_ERR_ARG, "row_index is NULL");
  return waic_impl(eng, ll, dtype, n_obs, row_index, n_rows, n_draws, stride_obs, stride_draw, scale_value, mem_space, stream,
                   lppd_i, var_i, waic_i, agg);
}

int pla_e_loo(pla_engine* eng, const void* x, const void* log_weights, const void* log_ratios, int dtype, int64_t n_obs,
              int64_t n_draws, int64_t stride_obs, int64_t stride_draw, int64_t tail_len, int mem_space, void* stream, double* mean,
              double* variance, double* k_mean, double* k_var, double* k_ratio) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (dtype != PLA_F64 && dtype != PLA_F32) return fail(PLA_ERR_ARG, "dtype must be PLA_F64 or PLA_F32");
  if (mem_space != PLA_HOST && mem_space != PLA_DEVICE) return fail(PLA_ERR_ARG, "bad mem_space");
  if (n_obs < 0) return fail(PLA_ERR_ARG, "n_obs < 0");
  if (n_obs > 0 && (!x || !log_weights)) return fail(PLA_ERR_ARG, "x / log_weights is NULL");
  if (n_draws < 1 || n_draws > (int64_t)1 << 30) return fail(PLA_ERR_ARG, "n_draws out of range");
  if (stride_draw <= 0 || stride_obs < 0) return fail(PLA_ERR_ARG, "bad strides");
  if (tail_len < 5) return fail(PLA_ERR_ARG, "tail_len must be at least 5");  // e_loo.py:298-299
  if (n_obs == 0) return PLA_OK;
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  // (row list of the one-pass wave kernel: the rows it leaves to the general one)
  if (int rc0 = grow(&eng->d_slow, &eng->d_slow_bytes, (size_t)n_obs * sizeof(unsigned))) return rc0;
  if (mem_space == PLA_DEVICE) {
    PLA_HIP(pla::launch_e_loo(x, log_weights, log_ratios, dtype, n_obs, (int)n_draws, stride_obs, stride_draw, (int)tail_len, mean,
                              variance, k_mean, k_var, k_ratio, (unsigned*)eng->d_slow, eng->counters, s));
    return PLA_OK;
  }
  // ---- PLA_HOST: row blocks of the two or three matrices through the staging buffers (d_in: x, d_lw: log-weights, d_slab: ratios)
  if (stride_draw != 1) return fail(PLA_ERR_UNSUPPORTED, "PLA_HOST input needs stride_draw == 1");
  const size_t esz = dtype == PLA_F64 ? 8 : 4;
  const size_t row_bytes = (size_t)n_draws * esz;
  int64_t rows_per_chunk = (int64_t)(((size_t)1 << 29) / row_bytes);
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  if (rows_per_chunk > n_obs) rows_per_chunk = n_obs;
  int rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * row_bytes);
  if (!rc) rc = grow(&eng->d_lw, &eng->d_lw_bytes, (size_t)rows_per_chunk * row_bytes);
  if (!rc && log_ratios) rc = grow(&eng->d_slab, &eng->d_slab_bytes, (size_t)rows_per_chunk * row_bytes);
  if (rc) return rc;
  {
    size_t have_b = eng->d_pw_elems * sizeof(double);
    rc = grow((void**)&eng->d_pw, &have_b, (size_t)(5 * rows_per_chunk) * sizeof(double));
    eng->d_pw_elems = have_b / sizeof(double);
    if (rc) return rc;
  }
  double* out[5] = {mean, variance, k_mean, k_var, k_ratio};
  for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
    const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
    const size_t off = (size_t)r0 * stride_obs * esz, pitch = (size_t)stride_obs * esz;
    PLA_HIP(hipMemcpy2DAsync(eng->d_in, row_bytes, (const char*)x + off, pitch, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
    PLA_HIP(hipMemcpy2DAsync(eng->d_lw, row_bytes, (const char*)log_weights + off, pitch, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
    if (log_ratios)
      PLA_HIP(hipMemcpy2DAsync(eng->d_slab, row_bytes, (const char*)log_ratios + off, pitch, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
    double* d = eng->d_pw;
    PLA_HIP(pla::launch_e_loo(eng->d_in, eng->d_lw, log_ratios ? eng->d_slab : nullptr, dtype, nr, (int)n_draws, n_draws, 1, (int)tail_len,
                              d, d + nr, d + 2 * nr, d + 3 * nr, d + 4 * nr, (unsigned*)eng->d_slow, eng->counters, s));
    for (int k = 0; k < 5; ++k)
      if (out[k]) PLA_HIP(hipMemcpyAsync(out[k] + r0, d + k * nr, (size_t)nr * sizeof(double), hipMemcpyDeviceToHost, s));
    PLA_HIP(hipStreamSynchronize(s));  // the staging buffers are reused by the next block
  }
  return PLA_OK;
}

int pla_e_loo_quantiles(pla_engine* eng, const void* x, const void* log_weights, int dtype, int64_t n_obs, int64_t n_draws,
                        int64_t stride_obs, int64_t stride_draw, const double* probs, int64_t n_probs, int mem_space, void* stream,
                        double* out) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (dtype != PLA_F64 && dtype != PLA_F32) return fail(PLA_ERR_ARG, "dtype must be PLA_F64 or PLA_F32");
  if (mem_space != PLA_HOST && mem_space != PLA_DEVICE) return fail(PLA_ERR_ARG, "bad mem_space");
  if (n_obs < 0 || n_probs < 0) return fail(PLA_ERR_ARG, "negative size");
  if (n_obs > 0 && (!x || !log_weights)) return fail(PLA_ERR_ARG, "x / log_weights is NULL");
  if (n_obs > 0 && n_probs > 0 && (!probs || !out)) return fail(PLA_ERR_ARG, "probs / out is NULL");
  if (n_draws < 1 || n_draws > (int64_t)1 << 30) return fail(PLA_ERR_ARG, "n_draws out of range");
  if (stride_draw <= 0 || stride_obs < 0) return fail(PLA_ERR_ARG, "bad strides");
  for (int64_t i = 0; i < n_probs; ++i)
    if (!(probs[i] > 0.0 && probs[i] < 1.0)) return fail(PLA_ERR_ARG, "probs must be between 0 and 1");  // e_loo.py:158-159
  if (n_obs == 0 || n_probs == 0) return PLA_OK;
  EngineCall call(eng, (hipStream_t)stream);
  PLA_HIP(hipSetDevice(eng->device));
  hipStream_t s = (hipStream_t)stream;
  int rc = grow(&eng->d_probs, &eng->d_probs_bytes, (size_t)n_probs * sizeof(double));
  if (rc) return rc;
  PLA_HIP(hipMemcpyAsync(eng->d_probs, probs, (size_t)n_probs * sizeof(double), hipMemcpyHostToDevice, s));
  PLA_HIP(hipStreamSynchronize(s));  // (the caller's probs array may go away after the call)
  const double* dp = (const double*)eng->d_probs;
  if (mem_space == PLA_DEVICE) {
    rc = grow(&eng->d_slow, &eng->d_slow_bytes, (size_t)n_obs * sizeof(unsigned));
    if (rc) return rc;
    PLA_HIP(pla::launch_e_loo_quantiles(x, log_weights, dtype, n_obs, (int)n_draws, stride_obs, stride_draw, dp, (int)n_probs, out,
                                        (unsigned*)eng->d_slow, eng->counters, s));
    return PLA_OK;
  }
  if (stride_draw != 1) return fail(PLA_ERR_UNSUPPORTED, "PLA_HOST input needs stride_draw == 1");
  const size_t esz = dtype == PLA_F64 ? 8 : 4;
  const size_t row_bytes = (size_t)n_draws * esz;
  int64_t rows_per_chunk = (int64_t)(((size_t)1 << 29) / row_bytes);
  if (rows_per_chunk < 1) rows_per_chunk = 1;
  if (rows_per_chunk > n_obs) rows_per_chunk = n_obs;
  rc = grow(&eng->d_in, &eng->d_in_bytes, (size_t)rows_per_chunk * row_bytes);
  if (!rc) rc = grow(&eng->d_lw, &eng->d_lw_bytes, (size_t)rows_per_chunk * row_bytes);
  if (rc) return rc;
  {
    size_t have_b = eng->d_pw_elems * sizeof(double);
    rc = grow((void**)&eng->d_pw, &have_b, (size_t)(rows_per_chunk * n_probs) * sizeof(double));
    eng->d_pw_elems = have_b / sizeof(double);
    if (rc) return rc;
  }
  rc = grow(&eng->d_slow, &eng->d_slow_bytes, (size_t)rows_per_chunk * sizeof(unsigned));
  if (rc) return rc;
  for (int64_t r0 = 0; r0 < n_obs; r0 += rows_per_chunk) {
    const int64_t nr = (n_obs - r0 < rows_per_chunk) ? (n_obs - r0) : rows_per_chunk;
    const size_t off = (size_t)r0 * stride_obs * esz, pitch = (size_t)stride_obs * esz;
    PLA_HIP(hipMemcpy2DAsync(eng->d_in, row_bytes, (const char*)x + off, pitch, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
    PLA_HIP(hipMemcpy2DAsync(eng->d_lw, row_bytes, (const char*)log_weights + off, pitch, row_bytes, (size_t)nr, hipMemcpyHostToDevice, s));
    PLA_HIP(pla::launch_e_loo_quantiles(eng->d_in, eng->d_lw, dtype, nr, (int)n_draws, n_draws, 1, dp, (int)n_probs, eng->d_pw,
                                        (unsigned*)eng->d_slow, eng->counters, s));
    PLA_HIP(hipMemcpyAsync(out + r0 * n_probs, eng->d_pw, (size_t)(nr * n_probs) * sizeof(double), hipMemcpyDeviceToHost, s));
    PLA_HIP(hipStreamSynchronize(s));
  }
  return PLA_OK;
}

int pla_fill_synthetic_chains(pla_engine* eng, void* ll_device, int dtype, int64_t n_obs, int64_t n_draws, int64_t row0,
                              uint64_t seed, int chains, double rho, double offset_sd, double k_lo, double k_hi, void* stream) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (dtype != PLA_F64 && dtype != PLA_F32) return fail(PLA_ERR_ARG, "dtype must be PLA_F64 or PLA_F32");
  if (n_obs < 0 || n_draws < 1) return fail(PLA_ERR_ARG, "bad shape");
  if (chains < 1 || chains > n_draws) return fail(PLA_ERR_ARG, "need 1 <= chains <= n_draws");
  if (!(rho > -1.0 && rho < 1.0) || !(offset_sd >= 0.0)) return fail(PLA_ERR_ARG, "need |rho| < 1 and offset_sd >= 0");
  if (n_obs > 0 && !ll_device) return fail(PLA_ERR_ARG, "ll_device is NULL");
  EngineCall call(eng);
  PLA_HIP(hipSetDevice(eng->device));
  PLA_HIP(pla::launch_fill_chains(ll_device, dtype, n_obs, n_draws, chains, rho, offset_sd, row0, seed, k_lo, k_hi, (hipStream_t)stream));
  return PLA_OK;
}

int pla_fill_synthetic(pla_engine* eng, void* ll_device, int dtype, int64_t n_obs, int64_t n_draws, int64_t row0,
                       uint64_t seed, double k_lo, double k_hi, double heavy_lo, double heavy_hi, void* stream) {
  if (!eng) return fail(PLA_ERR_ARG, "engine is NULL");
  if (dtype != PLA_F64 && dtype != PLA_F32) return fail(PLA_ERR_ARG, "dtype must be PLA_F64 or PLA_F32");
  if (n_obs < 0 || n_draws < 1) return fail(PLA_ERR_ARG, "bad shape");
  if (n_obs > 0 && !ll_device) return fail(PLA_ERR_ARG, "ll_device is NULL");
  EngineCall call(eng);
  PLA_HIP(hipSetDevice(eng->device));
  PLA_HIP(pla::launch_fill_synthetic(ll_device, dtype, n_obs, n_draws, row0, seed, k_lo, k_hi, heavy_lo, heavy_hi,
                                     (hipStream_t)stream));
  return PLA_OK;
}

}  // extern "C"
