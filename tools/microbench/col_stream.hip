// Read rate of an observations-fastest matrix (element (obs i, draw s) at in[s * ld + i]) in three access shapes; nothing is
// computed but a sum.  hipcc -O3 --offload-arch=gfx950 -o col_stream col_stream.hip ; ./col_stream [n_obs] [n_draws]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ld64(const __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  const v2i t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 2);
  return __hiloint2double(t[1], t[0]);
}
constexpr int U = 12;
// V0: one lane per observation, a wave reads 512 contiguous bytes of a draw
template <int THREADS>
__global__ __launch_bounds__(THREADS) void v0(const double* in, int64_t n_obs, int S, int64_t ld, double* out) {
  const int64_t i = (int64_t)blockIdx.x * THREADS + threadIdx.x;
  if (i >= n_obs) return;
  const int voff = (int)(i * 8);
  const int64_t db = ld * 8;
  double acc = 0.0, buf[2][U];
  const int nb = S / U;
  auto fetch = [&](double (&d)[U], int b) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)in + (int64_t)b * U * db, 0, (int)0xfffffff0u, 0x00020000);
#pragma unroll
    for (int u = 0; u < U; ++u) d[u] = ld64(rs, voff, (int)(u * db));
  };
  auto work = [&](const double (&d)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc += d[u];
  };
  fetch(buf[0], 0);
  int b = 0;
#pragma unroll 1
  for (; b + 2 <= nb; b += 2) {
    fetch(buf[1], b + 1);
    work(buf[0]);
    if (b + 2 < nb) fetch(buf[0], b + 2);
    work(buf[1]);
  }
  if (b < nb) work(buf[0]);
  out[i] = acc;
}
// V1: a workgroup owns G = 16 neighbouring observations; lane = (observation l & 15, draw l >> 4), wave w takes the draws
// 4 w .. 4 w + 3 of every step of 4 WAVES draws: one load instruction reads four 128-byte pieces, one per draw
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void v1(const double* in, int64_t n_obs, int S, int64_t ld, double* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int step = 4 * WAVES;
  const int64_t db = ld * 8;
  for (int64_t g = blockIdx.x; g < n_obs / 16; g += gridDim.x) {
    const int64_t i = g * 16 + (lane & 15);
    const int d0 = 4 * w + (lane >> 4);
    // per-lane offset: own observation + own draw of the step;  (d0 * ld * 8 must stay below 2^31: d0 < 32, ld <= 8e6)
    const int voff = (int)(i * 8 + d0 * db);
    double acc = 0.0, buf[2][U];
    const int nb = S / (U * step);
    auto fetch = [&](double (&d)[U], int b) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)in + (int64_t)b * U * step * db, 0, (int)0xfffffff0u, 0x00020000);
#pragma unroll
      for (int u = 0; u < U; ++u) d[u] = ld64(rs, voff, (int)(u * step * db));
    };
    auto work = [&](const double (&d)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += d[u];
    };
    fetch(buf[0], 0);
    int b = 0;
#pragma unroll 1
    for (; b + 2 <= nb; b += 2) {
      fetch(buf[1], b + 1);
      work(buf[0]);
      if (b + 2 < nb) fetch(buf[0], b + 2);
      work(buf[1]);
    }
    if (b < nb) work(buf[0]);
    if (acc == 12345.678) out[i] = acc;
  }
}
// V2: a workgroup of 4 waves owns 8 neighbouring observations (a 64-byte piece of a draw); lane = (observation l & 7, draw l >> 3).
// PAIR: the two halves of a 128-byte line go to two workgroups of the same XCD (blockIdx % 8) that run side by side.
template <bool PAIR>
__global__ __launch_bounds__(256) void v2(const double* in, int64_t n_obs, int S, int64_t ld, double* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int step = 32;
  const int64_t db = ld * 8;
  const int64_t ngroups = n_obs / 8;
  const int b = blockIdx.x, grid = gridDim.x;
  for (int64_t j = 0;; ++j) {
    int64_t g;
    if (PAIR) {
      const int xcd = b & 7, slot = b >> 3;
      g = 2 * ((int64_t)(slot >> 1) * 8 + xcd + j * (grid / 2)) + (slot & 1);
    } else {
      g = b + j * grid;
    }
    if (g >= ngroups) break;
    const int64_t i = g * 8 + (lane & 7);
    const int d0 = 8 * w + (lane >> 3);
    const unsigned voff = (unsigned)((i - g * 8) * 8) + (unsigned)(d0 * db);
    const char* gb = (const char*)in + g * 64;
    double acc = 0.0, buf[2][U];
    const int nb = S / (U * step);
    auto fetch = [&](double (&d)[U], int bb) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((char*)gb + (int64_t)(bb * U + u) * step * db, 0, (int)0xfffffff0u, 0x00020000);
        d[u] = ld64(rs, (int)voff, 0);
      }
    };
    auto work = [&](const double (&d)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += d[u];
    };
    fetch(buf[0], 0);
    int bb = 0;
#pragma unroll 1
    for (; bb + 2 <= nb; bb += 2) {
      fetch(buf[1], bb + 1);
      work(buf[0]);
      if (bb + 2 < nb) fetch(buf[0], bb + 2);
      work(buf[1]);
    }
    if (bb < nb) work(buf[0]);
    if (acc == 12345.678) out[i] = acc;
  }
}
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int S = argc > 2 ? atoi(argv[2]) : 4000;
  double* in; double* out;
  CK(hipMalloc(&in, (size_t)n * S * 8));
  CK(hipMalloc(&out, (size_t)n * 8));
  CK(hipMemset(in, 0x3f, (size_t)n * S * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-48s %8.3f ms  %6.2f TB/s\n", name, best, (double)n * S * 8 / (best * 1e-3) / 1e12);
    fflush(stdout);
  };
  // the bytes actually read: whole batches only (S / (U * step) * U * step draws)
  run("v0 lane per obs, 256 thr", [&] { hipLaunchKernelGGL(v0<256>, dim3((n + 255) / 256), dim3(256), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 4 waves, grid = groups", [&] { hipLaunchKernelGGL(v1<4>, dim3(n / 16), dim3(256), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 8 waves, grid = groups", [&] { hipLaunchKernelGGL(v1<8>, dim3(n / 16), dim3(512), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 8 waves, 256 persistent", [&] { hipLaunchKernelGGL(v1<8>, dim3(256), dim3(512), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 8 waves, 512 persistent", [&] { hipLaunchKernelGGL(v1<8>, dim3(512), dim3(512), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 4 waves, 512 persistent", [&] { hipLaunchKernelGGL(v1<4>, dim3(512), dim3(256), 0, 0, in, n, S, n, out); });
  run("v1 tile 16 obs, 4 waves, 1024 persistent", [&] { hipLaunchKernelGGL(v1<4>, dim3(1024), dim3(256), 0, 0, in, n, S, n, out); });

  run("v2 tile 8 obs (64 B pieces), 4 waves, 512 persistent", [&] { hipLaunchKernelGGL(v2<false>, dim3(512), dim3(256), 0, 0, in, n, S, n, out); });
  run("v2 tile 8 obs (64 B pieces), 4 waves, 512, XCD pairs", [&] { hipLaunchKernelGGL(v2<true>, dim3(512), dim3(256), 0, 0, in, n, S, n, out); });
  return 0;
}
