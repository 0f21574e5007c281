// Cost of LDS atomics with return per wave instruction, by the number of active lanes and the address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
// MODE 0: every lane its own counter; 1: lanes l & 15 share 16 counters (4 lanes each); 2: only lanes with (l % 10 == 0) active, own counters;
// 3: like 1 but only 1 lane in 10 adds to the shared counter, the others to their own; 4: ds_write_b64 all lanes; 5: ds_write_b64 1 lane in 10
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters) {
  __shared__ unsigned cnt[1024];
  __shared__ double slot[1024];
  const int tid = threadIdx.x, lane = tid & 63;
  cnt[tid] = 0; cnt[tid + 512] = 0;
  __syncthreads();
  unsigned acc = 0;
  unsigned* mine = &cnt[16 + tid];
  unsigned* shared_c = &cnt[lane & 15];
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool sel = ((lane + i + u) % 10) == 0;
      if (MODE == 0) acc += __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 1) acc += __hip_atomic_fetch_add(shared_c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) { if (sel) acc += __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (MODE == 3) acc += __hip_atomic_fetch_add(sel ? shared_c : mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 4) { slot[tid] = (double)(acc + i + u); acc += 1; }
      if (MODE == 5) { if (sel) slot[tid] = (double)(acc + i + u); acc += 1; }
      if (MODE == 6) { if (sel) acc += __hip_atomic_fetch_add(shared_c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
  }
  out[blockIdx.x * 512 + tid] = acc + (unsigned)slot[(tid * 7) & 1023];
}
int main() {
  unsigned* out; CK(hipMalloc(&out, 256 * 512 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  auto run = [&](const char* name, auto kern) {
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    // per CU: 8 waves x iters x 8 instructions
    printf("%-64s %8.3f ms  %6.1f ns per wave-instruction per CU (%.1f cycles at 2.2 GHz)\n", name, best, best * 1e6 / (8.0 * iters * 8), best * 1e6 / (8.0 * iters * 8) * 2.2);
  };
  run("0 atomic rtn, 64 lanes, own counters", k<0>);
  run("1 atomic rtn, 64 lanes, 16 shared counters (4 lanes each)", k<1>);
  run("2 atomic rtn, 1 lane in 10 active (exec mask), own counters", k<2>);
  run("3 atomic rtn, 64 lanes: 1 in 10 on shared counter, rest own", k<3>);
  run("6 atomic rtn, 1 lane in 10 active, shared counters", k<6>);
  run("4 ds_write_b64, 64 lanes", k<4>);
  run("5 ds_write_b64, 1 lane in 10 active", k<5>);
  return 0;
}
