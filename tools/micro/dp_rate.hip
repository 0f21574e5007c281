// Micro-benchmark: fp64 FMA issue rate on gfx950 as a function of independent chains per wave (ILP)
// and waves per SIMD (TLP).  hipcc --offload-arch=gfx950 -O3 dp_rate.hip -o dp_rate && ./dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void k(double* out, int iters, double a, double b) {
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = fma(x[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void k32(float* out, int iters, float a, float b) {
  float x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3f + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) x[c] = fmaf(x[c], a, b);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* d; hipMalloc(&d, 256 * 16 * 64 * 64 * sizeof(double));
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: blocks of 64 threads, 4*wps per CU
    const int blocks = 256 * 4 * wps;
#define RUN(CH) { float ms = timeit([&]{ hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0000001, 1e-9); }); \
    double instr = (double)iters * CH * wps; /* per SIMD */ \
    printf("f64 wps=%d chains=%d : %.2f cycles per wave-instr per SIMD (%.1f TFLOP/s)\n", wps, CH, ms * 1e-3 * 2.4e9 / instr, 2.0 * blocks * 64 * (double)iters * CH / (ms * 1e-3) / 1e12); }
    RUN(1) RUN(2) RUN(4) RUN(8)
#define RUN32(CH) { float ms = timeit([&]{ hipLaunchKernelGGL(k32<CH>, dim3(blocks), dim3(64), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f); }); \
    double instr = (double)iters * CH * wps; \
    printf("f32 wps=%d chains=%d : %.2f cycles per wave-instr per SIMD\n", wps, CH, ms * 1e-3 * 2.4e9 / instr); }
    RUN32(1) RUN32(4)
  }
  return 0;
}
