// Micro-benchmark: does a gfx950 SIMD co-issue VALU and SALU / LDS instructions of different waves?
// Each wave runs NV fp64 FMAs (8 independent chains) + NS scalar adds (+ NL LDS reads) per loop trip;
// 1, 2 and 4 waves per SIMD.  Also measures the core clock (s_memtime) against the 100 MHz
// real-time counter (s_memrealtime).   hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV, int NS, int NL>
__global__ __launch_bounds__(64) void mix(double* out, unsigned long long* clk, int iters, double a, double b) {
  __shared__ double lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
  __syncthreads();
  double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
  unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  double l0 = 0, l1 = 0;
  const unsigned laddr = (unsigned)(size_t)(lds + threadIdx.x) & 0xffffu;
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < NV) {
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
      }
      if (u < NS) {
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) : : "scc");
        asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) : : "scc");
      }
      if (u < NL) {
        asm volatile("ds_read_b64 %0, %1" : "=v"(l0) : "v"(laddr));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)(s0 + s1 + s2 + s3) + l0 + l1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  double* d; hipMalloc(&d, 256 * 16 * 64 * sizeof(double));
  unsigned long long* c; hipMalloc(&c, 16);
  const int iters = 4000;
#define RUN(NV, NS, NL) for (int wps : {1, 2, 4}) { const int blocks = 256 * 4 * wps; \
    float ms = timeit([&]{ hipLaunchKernelGGL((mix<NV, NS, NL>), dim3(blocks), dim3(64), 0, 0, d, c, iters, 1.0000001, 1e-9); }); \
    unsigned long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost); \
    printf("V=%2d S=%2d L=%d per trip, %d waves/SIMD: %.3f ms, %.1f ns per trip per SIMD-wave-set, memtime/realtime ticks %llu/%llu (ratio %.2f)\n", \
           NV * 8, NS * 8, NL, wps, ms, ms * 1e6 / iters, h[0], h[1], (double)h[0] / (double)h[1]); }
  RUN(8, 0, 0) RUN(0, 8, 0) RUN(8, 8, 0) RUN(8, 4, 0) RUN(8, 2, 0) RUN(8, 0, 4) RUN(8, 0, 8) RUN(8, 2, 4) RUN(4, 8, 0) RUN(0, 0, 8)
  return 0;
}
