// Device-side building blocks of the PSIS-LOO engine (gfx950 / CDNA4, wave64).
// Everything here is written for one workgroup == one observation (a row of S draws).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pla {

constexpr int kWave = 64;
constexpr double kLogTiny = -708.3964185322641;  // log(DBL_MIN): psis.py:90 / base.py:142
constexpr double kLn2 = 0.6931471805599453;
constexpr double kEps = 2.220446049250313e-16;   // np.finfo(float).eps
constexpr int kMaxGrid = 128;                    // m_est = 30 + isqrt(n) <= 120 for n <= 8192

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000ll); }
__device__ __forceinline__ double pinf() { return __longlong_as_double(0x7ff0000000000000ll); }

// ---- order-preserving 64-bit keys for doubles (no NaN expected) -------------------------
__device__ __forceinline__ uint64_t key_of(double x) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  if ((b << 1) == 0) b = 0;  // -0.0 and +0.0 compare equal in the reference
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(uint64_t k) {
  uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// ---- wave / workgroup reductions (result broadcast to every thread) ----------------------
struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = Op::f(v, __shfl_xor(v, o));
  return v;
}

// scratch: >= BLOCK/64 doubles of LDS. Two barriers (none for a one-wave workgroup).
template <class Op, int BLOCK>
__device__ __forceinline__ double block_reduce(double v, double* scratch) {
  v = wave_reduce<Op>(v);
  if constexpr (BLOCK > kWave) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    v = scratch[0];
#pragma unroll
    for (int i = 1; i < BLOCK / kWave; ++i) v = Op::f(v, scratch[i]);
  }
  return v;
}

// OR of small bit-flags across the workgroup (flags < 2^20, exact in double)
template <int BLOCK>
__device__ __forceinline__ unsigned block_or_bits(unsigned v, double* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= (unsigned)__shfl_xor((int)v, o);
  if constexpr (BLOCK > kWave) {
    unsigned* s = reinterpret_cast<unsigned*>(scratch);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[w] = v;
    __syncthreads();
    v = s[0];
#pragma unroll
    for (int i = 1; i < BLOCK / kWave; ++i) v |= s[i];
  }
  return v;
}

template <int BLOCK>
__device__ __forceinline__ void block_sync() {
  if constexpr (BLOCK > kWave) __syncthreads();
  else __builtin_amdgcn_wave_barrier();
}

// ---- product accumulator:  prod_i f_i  kept as mantissa * 2^exp ---------------------------
// sum_i log1p(-b*y_i) == log(prod_i (1 - b*y_i)); one log per product instead of one per factor.
struct ProdAcc {
  double m;
  int e;
  __device__ __forceinline__ void init() { m = 1.0; e = 0; }
  __device__ __forceinline__ void mul(double f) { m *= f; }
  __device__ __forceinline__ void renorm() {
    int ex;
    m = frexp(m, &ex);
    e += ex;
  }
  __device__ __forceinline__ double log_value() const { return log(m) + (double)e * kLn2; }
};

__device__ __forceinline__ int isqrt_i(int n) {
  int r = (int)sqrt((double)n);
  while (r * r > n) --r;
  while ((r + 1) * (r + 1) <= n) ++r;
  return r;
}

__device__ __forceinline__ int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// splitmix64 (Steele, Lea, Flood 2014): counter-based generator for the synthetic inputs
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ double u01_open(uint64_t r) {
  return ((double)(r >> 12) + 0.5) * (1.0 / 4503599627370496.0);  // strictly inside (0,1), exact
}

}  // namespace pla
