// Wave-per-observation LOO kernel (the production fast path for S <= 4096 draws, tail counts <= 250).
//
// One 64-lane wavefront owns one observation: the row of S draws sits in its registers (64 slots per
// lane, loaded once with 16-byte buffer loads), and every later step is done by that wave alone with
// DPP / permlane cross-lane reductions and a private LDS scratch.  Workgroups hold 4 such waves that only
// share read-only tables; there is NO workgroup barrier in the row loop, so the 8 waves resident on a CU
// run their phases (HBM load, exp sweep, selection, GPD fit) completely decoupled.
//
//   stats     max / min of the row and the smallest of the per-lane maxima over the first `gsz` slots
//             (one four-way reduction); a fixed-point bisection on ballots puts the speculative candidate
//             threshold t1 where ~2.2 (M+1) draws lie above it
//   sweep     for every draw ONE range reduction x = k*ln2/256 + r gives both e^x and e^-x: a 256-entry
//             LDS table of biased {2^(j/256), 2^(-j/256)} pairs, an even/odd polynomial and an integer
//             multiply-add into the exponent field.  Draws with x >= t1 are appended to an LDS list
//             (ballot + mbcnt rank, scalar running offset).  Behind the sweep the NEXT row's vectors
//             stream into the registers it has consumed.
//   select    512-bin histogram of the list over the integer keys k, DPP prefix scan -> boundary bin of
//             rank M; candidates at/above it are scattered to LDS grouped by bin and ranked exactly
//             inside their bin (4-wide batched LDS reads)
//   fit       Zhang-Stephens GPD fit, one lane per grid point b_j, log(prod(1 - b_j y_i)) over quads of
//             tail values (elementary symmetric sums) instead of m_est x n log1p; table-driven log
//   smooth    GPD quantiles from host tables, sums of the smoothed weights; loo_i / lppd_i from the sums
//
// Slots beyond S are padded so that no per-slot predicate is needed: first with a copy of the lane's own
// first draws (harmless for max / min), then with ll = -min raw (x = -R, the smallest x of the row),
// whose exactly known contribution is subtracted from the two sums.
//
// Rows the shortcuts cannot reproduce as the reference computes them (non-finite entries, more than 690
// nats of range, a threshold miss, a tail that cancels against the sum of all exponentials) are appended
// to a device list and recomputed by the general kernel (pla_rows.h).  Weights mode (LW = true) writes
// the normalised smoothed log-weights instead of loo_i / lppd_i.
#pragma once

#include <type_traits>

#include "pla_fast.h"
#include "pla_math.h"

#ifndef PLA_ROW_INLINE
#define PLA_ROW_INLINE __forceinline__
#endif

// total = (sum_all e^x - sum_tail e^x) + sum_tail w' cancels when the raw tail dominates the row; the sweep's exponentials
// carry up to 1.5e-13 relative (dropped r^4 term) that the tail's full-precision ones do not, so the result keeps
// ~1.5e-13 / guard: rows below the guard are left to the general kernel, which sums like the reference
#ifndef PLA_CANCEL_GUARD
#define PLA_CANCEL_GUARD 0.002
#endif
namespace pla {

constexpr double kCancelGuard = PLA_CANCEL_GUARD;
// split pass: 1 = the wave kernel hands the tail over grouped by bin and the fit kernel finishes the order (pla_fit.h);
// 0 = the wave kernel ranks the tail itself and hands it over ascending (round 1; kept for A/B runs)
#ifndef PLA_FIT_SORTS
#define PLA_FIT_SORTS 1
#endif
constexpr bool kFitSorts = PLA_FIT_SORTS != 0;
static_assert(PLA_FIT_SORTS == 1, "the hand-over is the tail's x grouped by bin (wave_select_split): the fit kernel sorts and exponentiates");

#ifndef PLA_WAVE_SLOTS
#define PLA_WAVE_SLOTS 64
#endif
#ifndef PLA_CAND_CAP
#define PLA_CAND_CAP 896
#endif
#ifndef PLA_WAVES_PER_BLOCK
#define PLA_WAVES_PER_BLOCK 4
#endif
#ifndef PLA_LOAD_AUX
#define PLA_LOAD_AUX 2  // cache policy of the single-read row loads: 2 = non-temporal
#endif
#ifndef PLA_LW_RESTREAM
#define PLA_LW_RESTREAM 0  // weights mode: 1 = the sweep reads its own row a second time (rounds 1-3), 0 = the row stays in its registers
#endif
#ifndef PLA_BISECT_ITERS
#define PLA_BISECT_ITERS 9
#endif
#ifndef PLA_SWEEP_DEPTH
#define PLA_SWEEP_DEPTH 3
#endif
#ifndef PLA_MIN_WAVES_PER_SIMD
#define PLA_MIN_WAVES_PER_SIMD 2
#endif
constexpr int kWaveSlots = PLA_WAVE_SLOTS;   // register slots per lane -> S <= 4096
constexpr int kWaveBins = 512;   // histogram bins over the candidate list
// LDS capacities of one wave's scratch, as a trait so that the kernel exists in two sizes
struct CapsSmall {                                  // S <= 4096, M <= 250: 8 waves per CU
  static constexpr int kCand = PLA_CAND_CAP;        // candidates (draws above the speculative threshold) kept in LDS
  static constexpr int kSa = 320;                   // candidates at/above the boundary bin: M + boundary-bin extras
  static constexpr int kMaxTail = 250;              // largest tail count M
  static constexpr int kWaves = PLA_WAVES_PER_BLOCK;  // independent waves per workgroup (they only share the tables)
};
struct CapsBig {                                    // long rows in chunks / small reff: M <= 512, 4 waves per CU
  static constexpr int kCand = 2048;
  static constexpr int kSa = 768;
  static constexpr int kMaxTail = 512;
  static constexpr int kWaves = 2;
};
struct CapsMid {                                    // chunked rows with M <= 448 (S = 20 000, reff = 1: C5): 8 waves per CU in the
  static constexpr int kCand = 1280;                // split pass, whose selection kernel keeps the exponential table only (the
  static constexpr int kSa = 512;                   // fused fallback, with its three more tables, runs one workgroup per CU)
  static constexpr int kMaxTail = 448;
  static constexpr int kWaves = 4;
};
// read-only tables of a kernel that only selects (split pass): the fit's tables stay with the fit kernel
struct WaveTabOnly {
  double tab[2 * kTabN];
};
struct CapsMidLW {                                  // weights mode of the chunked kernel: CapsMid's capacities + the candidates' draw indices,
  static constexpr int kCand = 1280;                // six waves in ONE workgroup per CU (156 KB of LDS with the four tables)
  static constexpr int kSa = 512;
  static constexpr int kMaxTail = 448;
  static constexpr int kWaves = 6;
};
struct CapsMid4 {                                   // M <= 320 (S up to ~11 000 at reff = 1, or reff >= 0.35 at S = 4000): 8 waves per CU
  static constexpr int kCand = 1088;
  static constexpr int kSa = 512;
  static constexpr int kMaxTail = 320;
  static constexpr int kWaves = 4;
};
constexpr int kCandCap = CapsSmall::kCand;
constexpr int kWaveCap = CapsSmall::kSa;
constexpr int kWaveMaxTail = CapsSmall::kMaxTail;
constexpr double kWaveMaxRange = 690.0;  // nats: e^x, e^-x and their sums over 4096 draws stay finite and normal

// ---- wave-wide reductions with DPP (result in SGPRs) -----------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov(double v, double ident) {
  const long long b = __double_as_longlong(v), o = __double_as_longlong(ident);
  const int lo = __builtin_amdgcn_update_dpp((int)(o & 0xffffffffll), (int)(b & 0xffffffffll), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(b >> 32), CTRL, ROWMASK, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double lane_value(double v, int lane) {  // wave-uniform result
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double uniform_d(double v) { return lane_value(v, 0); }

// One wavefront per workgroup: LDS operations of a wave are processed in order, so handing data
// between lanes through LDS needs no s_barrier -- and must not use __syncthreads(), whose fence
// drains the vector-memory queue (vmcnt(0)) and with it the next row's loads already in flight.
// The compiler only has to keep the LDS accesses in program order.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// max without the canonicalising self-max the compiler adds around fmax (NaNs are detected
// separately with v_cmp_class); NEG folds the sign flip of the first operand into the instruction
template <bool NEG>
__device__ __forceinline__ double vmax_nc(double a, double b) {
  double d;
  if constexpr (NEG) asm("v_max_f64 %0, -%1, %2" : "=v"(d) : "v"(a), "v"(b));
  else asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <bool NEG>
__device__ __forceinline__ float vmax_nc(float a, float b) {
  float d;
  if constexpr (NEG) asm("v_max_f32 %0, -%1, %2" : "=v"(d) : "v"(a), "v"(b));
  else asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// cross-lane move whose result is only meaningful in the lanes the control selects (rows masked out
// by ROWMASK keep whatever the destination register held)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_u(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double vmin_nc(double a, double b) {
  double d;
  asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float vmin_nc(float a, float b) {
  float d;
  asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// three-operand forms (32-bit only on this target): two slots of an f32 row per instruction in the row's max / min pass
template <bool NEG>
__device__ __forceinline__ float vmax3_nc(float a, float b, float c) {
  float d;
  if constexpr (NEG) asm("v_max3_f32 %0, -%1, -%2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  else asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float vmin3_nc(float a, float b, float c) {
  float d;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
template <RedOp OP>
__device__ __forceinline__ double red2(double a, double b) {
  if constexpr (OP == R_SUM) return a + b;
  else if constexpr (OP == R_MAX) return vmax_nc<false>(a, b);
  else return vmin_nc(a, b);
}
// Wave-wide reduction, result wave-uniform.  Three instructions per step: the last two steps
// (row_bcast) only produce valid data in the rows on the path to lane 63, which is the lane read.
// max / min do not canonicalise: NaNs are ignored (callers test finiteness separately).
template <RedOp OP>
__device__ __forceinline__ double wave_all(double v) {
  v = red2<OP>(v, dpp_mov_u<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = red2<OP>(v, dpp_mov_u<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = red2<OP>(v, dpp_mov_u<0x141, 0xF>(v));  // row_half_mirror
  v = red2<OP>(v, dpp_mov_u<0x140, 0xF>(v));  // row_mirror: every lane has its row's result
  v = red2<OP>(v, dpp_mov_u<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  v = red2<OP>(v, dpp_mov_u<0x143, 0xC>(v));  // row_bcast:31 into rows 2 and 3: lane 63 has it all
  return lane_value(v, 63);
}

// Two wave-wide reductions for the price of one and a bit: v_permlane32_swap (gfx950) leaves the pairwise
// combination of `a` across the two half-waves in lanes 0-31 and that of `b` in lanes 32-63; four DPP steps
// reduce inside the rows and one row_bcast:15 joins rows 0+1 (-> lane 31 = a) and rows 2+3 (-> lane 63 = b).
template <RedOp OP>
__device__ __forceinline__ void wave_all2(double a, double b, double& ra, double& rb) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  double v = red2<OP>(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
  v = red2<OP>(v, dpp_mov_u<0xB1, 0xF>(v));   // quad_perm [1,0,3,2]
  v = red2<OP>(v, dpp_mov_u<0x4E, 0xF>(v));   // quad_perm [2,3,0,1]
  v = red2<OP>(v, dpp_mov_u<0x141, 0xF>(v));  // row_half_mirror
  v = red2<OP>(v, dpp_mov_u<0x140, 0xF>(v));  // row_mirror
  v = red2<OP>(v, dpp_mov_u<0x142, 0xA>(v));  // row_bcast:15 into rows 1 and 3
  ra = lane_value(v, 31);
  rb = lane_value(v, 63);
}

// Four at once: after the half-wave swap of (a, b) and (c, d), v_permlane16_swap interleaves the rows so that
// each 16-lane row holds the partial results of ONE quantity; four DPP steps inside the rows finish the job
// (row 0 = a, row 1 = c, row 2 = b, row 3 = d).  21 + 8 instructions for four reductions.
template <RedOp OP>
__device__ __forceinline__ void wave_all4(double a, double b, double c, double d, double& ra, double& rb, double& rc, double& rd) {
  const auto l1 = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto h1 = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  const double ab = red2<OP>(__hiloint2double(h1[0], l1[0]), __hiloint2double(h1[1], l1[1]));  // rows 0,1: a; rows 2,3: b
  const auto l2 = __builtin_amdgcn_permlane32_swap(__double2loint(c), __double2loint(d), false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(__double2hiint(c), __double2hiint(d), false, false);
  const double cd = red2<OP>(__hiloint2double(h2[0], l2[0]), __hiloint2double(h2[1], l2[1]));  // rows 0,1: c; rows 2,3: d
  const auto l3 = __builtin_amdgcn_permlane16_swap(__double2loint(ab), __double2loint(cd), false, false);
  const auto h3 = __builtin_amdgcn_permlane16_swap(__double2hiint(ab), __double2hiint(cd), false, false);
  double v = red2<OP>(__hiloint2double(h3[0], l3[0]), __hiloint2double(h3[1], l3[1]));  // rows: a, c, b, d
  v = red2<OP>(v, dpp_mov_u<0xB1, 0xF>(v));
  v = red2<OP>(v, dpp_mov_u<0x4E, 0xF>(v));
  v = red2<OP>(v, dpp_mov_u<0x141, 0xF>(v));
  v = red2<OP>(v, dpp_mov_u<0x140, 0xF>(v));
  ra = lane_value(v, 0);
  rc = lane_value(v, 16);
  rb = lane_value(v, 32);
  rd = lane_value(v, 48);
}

template <class CAP>
struct WaveSmemT {
  using Caps = CAP;
  unsigned hist[kWaveBins];
  unsigned short start[kWaveBins];  // #candidates in bins above b
  double cand[CAP::kCand + 2 * kWave];  // candidate x values (+ 64 overflow slots + one dump slot per lane); reused for the
                                    // candidates sorted descending once they are binned
  double sa[CAP::kSa + 4];          // candidates at/above the boundary bin, grouped by bin (+ 4 sentinels); later y ascending
#if defined(PLA_PHASE_CLOCK)
  unsigned long long clk[8];        // diagnostic builds: cycles of this wave by phase (statistics incl. the wait for the row, threshold, sweep, selection) and rows
#endif
};
// weights mode (psislw): the candidates carry their draw index so that the smoothed tail can be
// written back to its positions
template <class CAP>
struct WaveSmemLWT : WaveSmemT<CAP> {
  unsigned short ids[CAP::kCand + 2 * kWave];  // draw index of cand[c]
  unsigned short sa_id[CAP::kSa + 4];          // ... of sa[c]
  unsigned short sb_id[CAP::kSa];              // ... of the sorted candidates
};
using WaveSmem = WaveSmemT<CapsSmall>;
using WaveSmemLW = WaveSmemLWT<CapsSmall>;
// read-only tables shared by the waves of a workgroup
template <class CAP>
struct WaveTablesT {
  double tab[2 * kTabN];            // biased {2^(j/256), 2^(-j/256)} pairs (pla_math.h): one 16-byte read serves both exponentials
  double lt[2 * kLogTabN];          // {1/c_j, log c_j} for log_tab (pla_math.h)
  double l1[CAP::kMaxTail + 6];      // log1p(-(j+0.5)/M), j < M  (host libm, psis.py:153,219-221)
  double bg[kWave];                 // 1 - sqrt(m_est/(j+0.5)) for m_est(M)  (psis.py:186)
};
using WaveTables = WaveTablesT<CapsSmall>;
constexpr int kWavesPerBlock = CapsSmall::kWaves;
__device__ __forceinline__ int wave_lane() { return (int)threadIdx.x & (kWave - 1); }

// phase ablation for profiling (tools/ablate.sh); compiled out of the production kernel
#ifndef PLA_WAVE_ABLATE
#define PLA_WAVE_ABLATE 0
#endif
// phase markers in the ISA for tools/isa_stats.py (comments only; off in the production build)
#if defined(PLA_PHASE_MARKS)
#define PLA_PHASE_STR2(n) #n
#define PLA_PHASE(n) asm volatile("; PLA_PHASE " PLA_PHASE_STR2(n))
#else
#define PLA_PHASE(n) ((void)0)
#endif

// a * b + c on the low 24 bits of a and b (one VALU op; b is a scalar register)
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
  int d;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
  return d;
}
// raw LDS byte address of an object in shared memory / store through one
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_store(unsigned addr, double v) {
  *(__attribute__((address_space(3))) double*)(uintptr_t)addr = v;
}
// (k & 255) << sh in one VALU op (SDWA byte select); sh lives in a VGPR
__device__ __forceinline__ unsigned byte0_shl(int k, int sh) {
  unsigned d;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0"
      : "=v"(d) : "v"(sh), "v"(k));
  return d;
}

// Issue the 16-byte buffer loads of one row into the register slots (slot q*VEC+e holds draw
// VEC*(lane + 64 q) + e).  The row is the bounds-checked range of the buffer descriptor: one VGPR
// offset (lane*16), scalar per-q offsets, zeros past the end.  Nothing waits here: the loads stay
// in flight until the slots are first used.
// AUX: cache policy of the loads (2 = non-temporal: the row is read once; 0 = default: weights mode reads
// the row a second time and wants it to stay in L2 / the Infinity Cache)
template <typename T, int VEC, int AUX = 2>
__device__ __forceinline__ void issue_row_loads(T (&v)[kWaveSlots], const T* rp, int S) {
  constexpr int NQ = kWaveSlots / VEC;
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int lane = wave_lane();
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(rp), 0, S * (int)sizeof(T), 0x00020000);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const v4i t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, q * (kWave * 16), AUX);
    if constexpr (VEC == 2) {
      v[2 * q] = (T)__hiloint2double(t[1], t[0]);
      v[2 * q + 1] = (T)__hiloint2double(t[3], t[2]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * q + e] = (T)__int_as_float(t[e]);
    }
  }
}

// one 16-byte vector (slots q*VEC .. q*VEC+VEC-1) of a row.  `sbase4k` (optional): a scalar register holding
// 4096 * (q / 4), so that four consecutive vectors share one scalar offset and differ in the instruction's immediate
// (0, 1024, 2048, 3072) instead of costing an s_movk each
template <typename T, int VEC, bool SHARED_BASE = false>
__device__ __forceinline__ void issue_row_vector(T (&v)[kWaveSlots], const __amdgpu_buffer_rsrc_t rs, int q, int sbase4k = 0) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i t;
  if constexpr (SHARED_BASE) t = __builtin_amdgcn_raw_buffer_load_b128(rs, wave_lane() * 16 + (q & 3) * (kWave * 16), sbase4k, PLA_LOAD_AUX);
  else t = __builtin_amdgcn_raw_buffer_load_b128(rs, wave_lane() * 16, q * (kWave * 16), PLA_LOAD_AUX);
  if constexpr (VEC == 2) {
    v[2 * q] = (T)__hiloint2double(t[1], t[0]);
    v[2 * q + 1] = (T)__hiloint2double(t[3], t[2]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = (T)__int_as_float(t[e]);
  }
}

// Overwrite the slots of vectors q >= qfull that lie past the end of the row (lanes >= qrem of
// vector qfull, every lane of the later ones) with `padv` (CONST) or with the lane's first vector.
// Written as a recursion from the last vector down so that it compiles to a chain of wave-uniform
// early exits (usually one vector is partial) rather than NQ predicated selects.
template <typename T, int VEC, int Q, bool CONST>
__device__ __forceinline__ void pad_tail(T (&v)[kWaveSlots], int qfull, int qrem, T padv) {
  if constexpr (Q >= 1) {
    if (Q >= qfull) {
      asm volatile("");  // keep the branch
      const bool ok = (Q == qfull) && (wave_lane() < qrem);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[Q * VEC + e] = ok ? v[Q * VEC + e] : (CONST ? padv : v[e]);
      pad_tail<T, VEC, Q - 1, CONST>(v, qfull, qrem, padv);
    }
  }
}


// Per-lane maximum and minimum of raw (= -v in LOO mode, v in weights mode) over the 64 slots, and the maximum over the first
// `gsz` slots in visiting order (the threshold sample).  The order is a compile-time permutation of the vectors
// (bitrev_order over 2^B of them); `bits` picks among the instantiations with a wave-uniform branch, so the pass stays
// two VALU operations per slot.
template <typename T, int VEC, bool LW, int B>
__device__ __forceinline__ void row_stats_b(const T (&v)[kWaveSlots], const int gsz, double& mx, double& mn, double& gs) {
  // two accumulators per quantity: a dependent fp64 VALU pair needs a wait state the compiler fills with s_nop when the
  // chain is only two instructions long; four interleaved chains need none
  const T ninf = (T)(-pinf());
  T cur[2] = {ninf, ninf}, vmx[2] = {LW ? (T)pinf() : ninf, LW ? (T)pinf() : ninf}, snap = ninf;
  if constexpr (sizeof(T) == 4) {
    // f32 rows: v_max3_f32 / v_min3_f32 take two slots at a time -- one VALU operation per slot instead of two (the group
    // boundaries 4, 8, 16, 32 fall between pairs; the slots of a pair lie in one 16-byte vector)
    static_assert(VEC % 2 == 0, "pairs of slots inside a vector");
#pragma unroll
    for (int i = 0; i < kWaveSlots; i += 2) {
      const int s0 = bitrev_order(i / VEC, B) * VEC + i % VEC, s1 = s0 + 1, a = (i >> 1) & 1;
      cur[a] = vmax3_nc<!LW>(v[s0], v[s1], cur[a]);
      vmx[a] = LW ? vmin3_nc(v[s0], v[s1], vmx[a]) : vmax3_nc<false>(v[s0], v[s1], vmx[a]);
      if (i == 2 || i == 6 || i == 14 || i == 30) {
        if (gsz == i + 2) {
          asm volatile("");
          snap = vmax_nc<false>(cur[0], cur[1]);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < kWaveSlots; ++i) {
      const int sl = bitrev_order(i / VEC, B) * VEC + i % VEC;
      cur[i & 1] = vmax_nc<!LW>(v[sl], cur[i & 1]);                                   // max raw
      vmx[i & 1] = LW ? vmin_nc(v[sl], vmx[i & 1]) : vmax_nc<false>(v[sl], vmx[i & 1]);  // min raw (LOO: as max ll)
      if (i == 3 || i == 7 || i == 15 || i == 31) {  // gsz is one of 4, 8, 16, 32 (wave_threshold_params)
        if (gsz == i + 1) {
          asm volatile("");  // a real wave-uniform branch, not a select per slot
          snap = vmax_nc<false>(cur[0], cur[1]);
        }
      }
    }
  }
  mx = (double)vmax_nc<false>(cur[0], cur[1]);
  const T vm = LW ? vmin_nc(vmx[0], vmx[1]) : vmax_nc<false>(vmx[0], vmx[1]);
  mn = LW ? (double)vm : -(double)vm;
  gs = (double)snap;
}
template <typename T, int VEC, bool LW>
__device__ __forceinline__ void row_stats(const T (&v)[kWaveSlots], const int gsz, const int bits, double& mx, double& mn, double& gs) {
  constexpr int NQ = kWaveSlots / VEC;
  constexpr int L = NQ == 32 ? 5 : (NQ == 16 ? 4 : 3);
  if (bits == L) row_stats_b<T, VEC, LW, L>(v, gsz, mx, mn, gs);
  else if (bits == L - 1) row_stats_b<T, VEC, LW, L - 1>(v, gsz, mx, mn, gs);
  else if (bits == L - 2) row_stats_b<T, VEC, LW, L - 2>(v, gsz, mx, mn, gs);
  else row_stats_b<T, VEC, LW, 0>(v, gsz, mx, mn, gs);
  // every row load has landed by now (each order ends on the last vector).  Said explicitly, because the compiler's
  // wait-count bookkeeping loses track across the merge of the four variants and would otherwise guard every later
  // read of the row with a (satisfied, but issued) s_waitcnt vmcnt.
#ifndef PLA_STATS_VMCNT
#define PLA_STATS_VMCNT 0x0F70  // vmcnt(0), expcnt / lgkmcnt untouched
#endif
#if PLA_STATS_VMCNT
  __builtin_amdgcn_s_waitcnt(PLA_STATS_VMCNT);
#endif
}

// Number of draws of the register block at or above a raw threshold (pads hold the row minimum by then and never count).  `thr` is in the sign convention of the stored values
// (LOO mode keeps ll = -raw).  Wave-uniform result.
__device__ __forceinline__ int wave_sum_int(int v) {  // wave-uniform sum of a per-lane integer
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, false);  // row_mirror: every lane has its row's sum
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
         __builtin_amdgcn_readlane(v, 48);
}
// cnt += (GE ? a >= b : a <= b), per lane: a compare into vcc and an add-with-carry -- written out because the compiler would
// first collect the 64 masks of an unrolled count in scalar registers and spill them
template <bool GE>
__device__ __forceinline__ void count_if(int& cnt, double a, double b) {
  if constexpr (GE) asm volatile("v_cmp_ge_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a), "v"(b) : "vcc");
  else asm volatile("v_cmp_le_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a), "v"(b) : "vcc");
}
template <bool GE>
__device__ __forceinline__ void count_if(int& cnt, float a, float b) {
  if constexpr (GE) asm volatile("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a), "v"(b) : "vcc");
  else asm volatile("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(cnt) : "v"(a), "v"(b) : "vcc");
}
// (Weights mode rewrites its pads late -- doing it early costs that kernel registers it does not have -- so there the
// slots past the row still hold copies of the lane's first draws and are counted with them: exact to 96 / 4096 at S = 4000,
// cruder for rows much shorter than the register block; the candidate count after the sweep is exact either way.)
// four slots at a time, counted on the SCALAR side: four compares into four scalar masks, their population counts and the
// running sum are scalar instructions -- one vector operation per slot instead of two (the pass is bound by vector issue), and
// the wave-uniform total needs no reduction across the lanes.  (Left to the compiler, the 64 masks of the unrolled count are all
// computed first and spilled: hence four per block, written out.)
#ifndef PLA_COUNT_SALU
#define PLA_COUNT_SALU 1
#endif
template <bool GE>
__device__ __forceinline__ void count4_scalar(int& cnt, double a0, double a1, double a2, double a3, double b) {
  unsigned long long m0, m1, m2, m3;
  int c0, c1, c2, c3;
  if constexpr (GE)
    asm volatile("v_cmp_ge_f64 %1, %9, %13\n\tv_cmp_ge_f64 %2, %10, %13\n\tv_cmp_ge_f64 %3, %11, %13\n\tv_cmp_ge_f64 %4, %12, %13\n\t"
                 "s_bcnt1_i32_b64 %5, %1\n\ts_bcnt1_i32_b64 %6, %2\n\ts_bcnt1_i32_b64 %7, %3\n\ts_bcnt1_i32_b64 %8, %4\n\t"
                 "s_add_i32 %5, %5, %6\n\ts_add_i32 %7, %7, %8\n\ts_add_i32 %0, %0, %5\n\ts_add_i32 %0, %0, %7"
                 : "+s"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b)
                 : "scc");
  else
    asm volatile("v_cmp_le_f64 %1, %9, %13\n\tv_cmp_le_f64 %2, %10, %13\n\tv_cmp_le_f64 %3, %11, %13\n\tv_cmp_le_f64 %4, %12, %13\n\t"
                 "s_bcnt1_i32_b64 %5, %1\n\ts_bcnt1_i32_b64 %6, %2\n\ts_bcnt1_i32_b64 %7, %3\n\ts_bcnt1_i32_b64 %8, %4\n\t"
                 "s_add_i32 %5, %5, %6\n\ts_add_i32 %7, %7, %8\n\ts_add_i32 %0, %0, %5\n\ts_add_i32 %0, %0, %7"
                 : "+s"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b)
                 : "scc");
}
template <bool GE>
__device__ __forceinline__ void count4_scalar(int& cnt, float a0, float a1, float a2, float a3, float b) {
  unsigned long long m0, m1, m2, m3;
  int c0, c1, c2, c3;
  if constexpr (GE)
    asm volatile("v_cmp_ge_f32 %1, %9, %13\n\tv_cmp_ge_f32 %2, %10, %13\n\tv_cmp_ge_f32 %3, %11, %13\n\tv_cmp_ge_f32 %4, %12, %13\n\t"
                 "s_bcnt1_i32_b64 %5, %1\n\ts_bcnt1_i32_b64 %6, %2\n\ts_bcnt1_i32_b64 %7, %3\n\ts_bcnt1_i32_b64 %8, %4\n\t"
                 "s_add_i32 %5, %5, %6\n\ts_add_i32 %7, %7, %8\n\ts_add_i32 %0, %0, %5\n\ts_add_i32 %0, %0, %7"
                 : "+s"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b)
                 : "scc");
  else
    asm volatile("v_cmp_le_f32 %1, %9, %13\n\tv_cmp_le_f32 %2, %10, %13\n\tv_cmp_le_f32 %3, %11, %13\n\tv_cmp_le_f32 %4, %12, %13\n\t"
                 "s_bcnt1_i32_b64 %5, %1\n\ts_bcnt1_i32_b64 %6, %2\n\ts_bcnt1_i32_b64 %7, %3\n\ts_bcnt1_i32_b64 %8, %4\n\t"
                 "s_add_i32 %5, %5, %6\n\ts_add_i32 %7, %7, %8\n\ts_add_i32 %0, %0, %5\n\ts_add_i32 %0, %0, %7"
                 : "+s"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b)
                 : "scc");
}
template <typename T, bool LW>
__device__ __forceinline__ int count_above(const T (&v)[kWaveSlots], T thr) {
  asm volatile("" : "+v"(thr));
#if PLA_COUNT_SALU
  int total = 0;  // wave-uniform, in a scalar register
#pragma unroll
  for (int i = 0; i < kWaveSlots; i += 4) count4_scalar<LW>(total, v[i], v[i + 1], v[i + 2], v[i + 3], thr);
  return total;
#else
  int cnt = 0;  // per lane
#pragma unroll
  for (int i = 0; i < kWaveSlots; ++i) count_if<LW>(cnt, v[i], thr);
  return wave_sum_int(cnt);
#endif
}
// The speculative threshold comes from an order statistic of per-lane GROUP maxima over a sample of the row: a biased
// quantile estimate when the draws of a group are dependent (autocorrelated MCMC output: the sample is made of 128-draw
// blocks) and meaningless for rows that trend or are sorted.  So it is checked against what it is meant to deliver -- the
// EXACT number of draws of the register block at or above it (two VALU operations per slot) -- and when that count is
// outside [cr_lo, cr_hi] the threshold is replaced by one from bisection on such counts (a handful of steps): any order of
// the draws stays on the fast path.  (A count over the sample alone was tried first: under AR(1) with rho = 0.9 it let 3 %
// of the rows through with twice the candidates the list holds, and sent descending rows searching on the wrong side.)
// `t_raw`: threshold on the raw scale, replaced when a better one is found; false: none found.
template <typename T, int VEC, bool LW>
__device__ __forceinline__ bool wave_threshold_check(const T (&v)[kWaveSlots], const FastParams& F, const double raw_min,
                                                     const double raw_max, double& t_raw) {
  if (__builtin_amdgcn_readfirstlane(F.cr_hi) <= 0) return true;
  const auto stored = [](double raw) { return LW ? (T)raw : (T)(-raw); };
  // (counting every other vector only, with the band scaled to that half, was tried in round 4: -0.4 % on iid rows, but the
  // half that is not counted differs enough on chain-major AR(1) rows to send 4 rows in 10 000 to the general kernel)
  const auto count = [&](T thr) { return count_above<T, LW>(v, thr); };
  const int c0 = count(stored(t_raw));
  if (c0 >= __builtin_amdgcn_readfirstlane(F.cr_lo) && c0 <= __builtin_amdgcn_readfirstlane(F.cr_hi)) return true;
  // (the bracket lives in vector registers: the callers' row loops are short of scalar ones)
  double lo = (c0 > __builtin_amdgcn_readfirstlane(F.cr_hi)) ? t_raw : raw_min;
  double hi = (c0 > __builtin_amdgcn_readfirstlane(F.cr_hi)) ? raw_max : t_raw;
  asm volatile("" : "+v"(lo), "+v"(hi));
#pragma unroll 1
  for (int it = 0; it < 24; ++it) {
    double mid = 0.5 * (lo + hi);
    asm volatile("" : "+v"(mid));
    const int c = count(stored(mid));
    if (c > __builtin_amdgcn_readfirstlane(F.cr_hi)) lo = mid;
    else if (c < __builtin_amdgcn_readfirstlane(F.cr_lo)) hi = mid;
    else {
      t_raw = uniform_d(mid);
      return true;
    }
    asm volatile("" : "+v"(lo), "+v"(hi));
  }
  return false;  // (e.g. a block of tied draws straddles the whole band)
}

// Everything after the sweep: exact selection of the M+1 largest among the candidates, GPD fit,
// smoothing sums and the outputs.  Shared by the one-chunk and the chunked front ends; `lppd_shift` is
// the log of the factor by which the chunked front's s2 is short (0 otherwise).
// ---- weights mode, output: lw_s = (raw_s - m) - L for the (up to) 4096 draws held in the row registers, 16 bytes per lane
// and store (stores past the end of `ro` are dropped).  STREAM: once a vector is stored, the same registers take the vector of
// the next chunk from `rs_next` (chunked kernel: the row passes through the registers a second time).
// CLAMP: lw_s = min(raw_s - m, xcap) - L (truncated importance sampling, tis.py:107-114).
template <typename T, int VEC, bool STREAM, bool CLAMP = false>
__device__ __forceinline__ void lw_store_chunk(T (&v)[kWaveSlots], const __amdgpu_buffer_rsrc_t ro, const __amdgpu_buffer_rsrc_t rs_next,
                                               const int lane, const int qfull, const double m, const double L,
                                               const double xcap = 0.0) {
  constexpr int NQ = kWaveSlots / VEC;
  typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q <= qfull) {  // later vectors lie past the row (stores past the end are dropped anyway)
      v4i t;
      if constexpr (VEC == 2) {
        double x0 = (double)v[2 * q] - m, x1 = (double)v[2 * q + 1] - m;
        if constexpr (CLAMP) { x0 = fmin(x0, xcap); x1 = fmin(x1, xcap); }
        const double a0 = x0 - L, a1 = x1 - L;
        t[0] = __double2loint(a0); t[1] = __double2hiint(a0);
        t[2] = __double2loint(a1); t[3] = __double2hiint(a1);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double xe = (double)v[4 * q + e] - m;
          if constexpr (CLAMP) xe = fmin(xe, xcap);
          t[e] = __float_as_int((float)(xe - L));
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(t, ro, lane * 16, q * (kWave * 16), 0);  // (non-temporal stores measured 5 % slower)
      // gfx9 hazard: a VALU write to the data registers of a > 8-byte buffer store with an SGPR offset
      // needs a wait state after the store.  The compiler's hazard pass misses it across the block
      // boundary that follows the last store (observed: the low dword of the stored value replaced
      // by the next instruction's literal), so the registers are kept alive over one s_nop.
      asm volatile("s_nop 0" : : "v"(t));
    }
    if constexpr (STREAM) issue_row_vector<T, VEC>(v, rs_next, q);
  }
}
// the smoothed tail at its positions (psis.py:155-158); must land after the row's stores (waiting for them to retire costs
// nothing measurable: issuing the patch right behind them -- same wave, same addresses -- 7.27 against 7.27 ms)
template <typename T, class SM, class TB>
__device__ __forceinline__ void lw_patch_tail(SM& sm, const TB& tb, T* orow, const int lane, const int n, const double L) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const double* wtail = sm.cand + SM::Caps::kSa + kWave;
  for (int j = lane; j < n; j += kWave) {
    const int id = (int)sm.sb_id[n - 1 - j];
    orow[id] = (T)(log_tab(wtail[j], tb.lt) - L);
  }
}

// (DEFER, weights mode of the chunked kernel: the row is not in the registers, so nothing is stored here; `loo` returns
// L = log(total) and `lppd` the smoothed tail length (0: nothing to patch) for lw_store_chunk / lw_patch_tail below)
// (the split pass never comes here: its selection is wave_select_split() below, the ONE producer of the hand-over's format)
template <typename T, int VEC, bool LW, typename SM, typename TB, bool DEFER = false>
__device__ __forceinline__ void wave_back(const RowsParams& P, SM& sm, const TB& tb, const int64_t r, T (&v)[kWaveSlots],
                                          const int lane, const int S, const int M, const int mestM, const double logS,
                                          const int dbgs, const double m, const double mn, const double R,
                                          const double lppd_shift, double s1, double s2, const unsigned ncand,
                                          const int k1, const int sh, const double magic, const double c256,
                                          const int qfull, const int qrem, bool& slow, double& khat, double& loo,
                                          double& lppd) {
  constexpr int NQ = kWaveSlots / VEC;
  constexpr int kSa = SM::Caps::kSa;
  const double INF = pinf();
  const double* l1tab = tb.l1;
  const double* bgrid = tb.bg;
  const auto key_of = [&](double xx) { return __double2loint(fma(xx, c256, magic)); };
  (void)NQ;
  // ---- 3. histogram of the candidate list, suffix scan (8 bins per lane) ---------------------
  PLA_PHASE(4);
  unsigned one = 1u;
  asm volatile("" : "+v"(one));
  // four list entries per lane and trip: the LDS reads go out together instead of one round trip each
  for (unsigned c0 = lane; c0 < ncand; c0 += 4 * kWave) {
    double xs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned c = c0 + u * kWave;
      xs[u] = sm.cand[c < ncand ? c : c0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned c = c0 + u * kWave;
      atomicAdd(&sm.hist[(key_of(xs[u]) - k1) >> sh], c < ncand ? one : 0u);  // past the end: add 0
    }
  }
  wave_sync();
  PLA_PHASE(5);
  int bstar = 0, C1 = 0, nbnd = 0;  // boundary bin, candidates at/above it, candidates IN it
  {
    unsigned c[8];
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 h = *reinterpret_cast<const uint4*>(&sm.hist[8 * lane + 4 * i]);
      c[4 * i] = h.x; c[4 * i + 1] = h.y; c[4 * i + 2] = h.z; c[4 * i + 3] = h.w;
      tot += h.x + h.y + h.z + h.w;
    }
    // inclusive prefix sum over the 64 lanes with DPP adds (no LDS round trips), then suffix = total - prefix
    unsigned pre = tot;
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x111, 0xF, 0xF, true);   // row_shr:1
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x112, 0xF, 0xF, true);   // row_shr:2
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x113, 0xF, 0xF, true);   // row_shr:3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x114, 0xF, 0xE, false);  // row_shr:4, banks 1-3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x118, 0xF, 0xC, false);  // row_shr:8, banks 2-3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x142, 0xA, 0xF, false);  // row_bcast:15
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x143, 0xC, 0xF, false);  // row_bcast:31
    const unsigned all = (unsigned)__builtin_amdgcn_readlane((int)pre, kWave - 1);
    const unsigned suf = all - pre + tot;  // this lane's bins and everything above
    unsigned a = suf - tot;  // candidates in bins owned by higher lanes
    int fb = -1, fc = 0, fn = 0;
    unsigned st[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      st[i] = a;
      if ((unsigned)M >= a && (unsigned)M < a + c[i]) {
        fb = 8 * lane + i;
        fc = (int)(a + c[i]);
        fn = (int)c[i];
      }
      a += c[i];
    }
    *reinterpret_cast<uint4*>(&sm.start[8 * lane]) =
        make_uint4(st[0] | (st[1] << 16), st[2] | (st[3] << 16), st[4] | (st[5] << 16), st[6] | (st[7] << 16));
    const unsigned long long who = __ballot(fb >= 0);
    const int src = __ffsll((long long)who) - 1;
    bstar = __builtin_amdgcn_readlane(fb, src);
    C1 = __builtin_amdgcn_readlane(fc, src);
    nbnd = __builtin_amdgcn_readlane(fn, src);
  }
  wave_sync();
  if (C1 > kSa) {
    slow = true;
  } else {
    // ---- 4. candidates at/above the boundary bin -> sa, grouped by bin (descending bins) --------
    PLA_PHASE(6);
    const int kstar = k1 + (bstar << sh);
    for (unsigned c0 = lane; c0 < ncand; c0 += 4 * kWave) {
      double xs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned c = c0 + u * kWave;
        xs[u] = sm.cand[c < ncand ? c : c0];
      }
      unsigned short idv[4] = {0, 0, 0, 0};
      if constexpr (LW) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned c = c0 + u * kWave;
          idv[u] = sm.ids[c < ncand ? c : c0];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned c = c0 + u * kWave;
        const double x = xs[u];
        const int k = key_of(x);
        if (c < ncand && k >= kstar) {
          const int b = (k - k1) >> sh;
          const unsigned slot = sm.start[b] + (atomicSub(&sm.hist[b], 1u) - 1u);
          sm.sa[slot] = x;
          if constexpr (LW) sm.sa_id[slot] = idv[u];
        }
      }
    }
    if (lane < 4) sm.sa[C1 + lane] = -INF;  // sentinels for the 4-wide reads of the ranking loop
    wave_sync();
    PLA_PHASE(7);
    double* sb = sm.cand;  // the list is consumed: its storage now holds the sorted candidates
    // ---- 5. exact descending rank inside each bin (ties: arbitrary, the sums do not care) --
    for (int c = lane; c < C1; c += kWave) {
      const double x = sm.sa[c];
      const int b = (key_of(x) - k1) >> sh;
      const int lo = (int)sm.start[b];
      const int hi = (b > 0) ? (int)sm.start[b - 1] : C1;
      int cnt = 0, same = 0;
      // four neighbours per trip (one LDS round trip): entries past the bin are smaller values of
      // lower bins, entries past the end are -inf sentinels, so neither counts
      for (int c2 = lo; c2 < hi; c2 += 4) {
        double x2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x2[u] = sm.sa[c2 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          cnt += (x2[u] > x) ? 1 : 0;
          same += (x2[u] == x) ? 1 : 0;  // counts the element itself
        }
      }
      if (__ballot(same > 1) != 0ull) {  // duplicates (repeated draws): order them by position
        if (same > 1)
          for (int c2 = lo; c2 < c; ++c2) cnt += (sm.sa[c2] == x) ? 1 : 0;
      }
      sb[lo + cnt] = x;
      if constexpr (LW) sm.sb_id[lo + cnt] = sm.sa_id[c];
    }
    wave_sync();
    // ---- cutoff (psis.py:135-141); R < 690: the log(DBL_MIN) floor cannot bind -------------
    PLA_PHASE(8);
    const double xcut = sb[M];
    int n = M;
    while (n > 0 && sb[n - 1] == xcut) --n;  // ties at the cutoff leave the tail
    const double e_cut = exp_tab(xcut, tb.tab);
    double acc_t = 0.0, acc_r = 0.0;  // (sum w' - sum e) and sum w'/e over the tail
    bool smoothed = false;
    if (n > 4 && !(dbgs & 8)) {
      wave_sync();
      // y ascending (psis.py:146-147), stored with the pair sums / products the fit loop eats
      // (the reciprocals e^-x of the same range reduction are kept for the weight ratios below)
      double* inv_e = sb + kSa + kWave;  // beyond the sorted candidates and the pair data
      const int n32 = (n + 31) & ~31;  // padded with y = 0 (factor 1) so that the fit runs whole trips only
      // Straight-line code for 3 (n <= 192, the usual case) or 4 elements per lane: the independent
      // exp chains interleave instead of running one LDS round trip + ~20 dependent fp64 ops at a time.
      // Every slot up to 64 U is written (zeros past n), so the later loops need no bounds either.
      const bool three = n32 <= 3 * kWave;
      const auto y_pass = [&](auto UC) {
        constexpr int U = decltype(UC)::value;
        double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = lane + kWave * u;
          xv[u] = sb[j < n ? n - 1 - j : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = lane + kWave * u;
          double ep, en;
          exp_pair(xv[u], tb.tab, ep, en);
          sm.sa[j] = j < n ? ep - e_cut : 0.0;
          inv_e[j] = en;
        }
      };
      constexpr bool kBigTail = SM::Caps::kMaxTail > 4 * kWave;  // tails of up to 8 elements per lane
      const bool eight = kBigTail && n32 > 4 * kWave;
      if (three) y_pass(std::integral_constant<int, 3>{});
      else if (!eight) y_pass(std::integral_constant<int, 4>{});
      else if constexpr (kBigTail) y_pass(std::integral_constant<int, 8>{});
      wave_sync();
      const double* y = sm.sa;
      const double nn = (double)n;
      // prod_{i<4} (1 - b y_i) = 1 - e1 b + e2 b^2 - e3 b^3 + e4 b^4: the elementary symmetric sums of every
      // four consecutive y, once per row (4 fma + 1 mul per grid point and quad instead of 2 x 3 for pairs)
      {
        constexpr int QU = kBigTail ? 2 : 1;  // quads per lane
        double2 ya[QU], yb[QU];
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          ya[u] = *reinterpret_cast<const double2*>(y + 4 * (lane + kWave * u));
          yb[u] = *reinterpret_cast<const double2*>(y + 4 * (lane + kWave * u) + 2);
        }
#pragma unroll
        for (int u = 0; u < QU; ++u) {
          const double s01 = ya[u].x + ya[u].y, p01 = ya[u].x * ya[u].y, s23 = yb[u].x + yb[u].y, p23 = yb[u].x * yb[u].y;
          double* o = &sb[4 * (lane + kWave * u)];
          *reinterpret_cast<double2*>(o) = make_double2(s01 + s23, fma(s01, s23, p01 + p23));
          *reinterpret_cast<double2*>(o + 2) = make_double2(fma(p01, s23, p23 * s01), p01 * p23);
        }
      }
      wave_sync();
      const double* yp = sb;
      // ---- 6. GPD fit (psis.py:163-208), lane j <-> grid point b_j -------------------------
      PLA_PHASE(9);
      const int mest = 30 + isqrt_i(n);
      const double yq = y[((n + 2) >> 2) - 1];
      const double yn = y[n - 1];
      const bool act = lane < mest;
      // psis.py:186: 1 - sqrt(m_est / (j - 0.5)); host table for the usual n == M
      double b = (mest == mestM) ? bgrid[lane] : 1.0 - sqrt((double)mest / ((double)(lane + 1) - 0.5));
      b = div_fast(b, 3.0 * yq);   // psis.py:187
      b += recip_fast(yn);         // psis.py:188
      const double b_first = lane_value(b, 0);  // most negative grid point
      const double b_last = lane_value(b, mest - 1);
      const double fbig = fma(-b_first, yn, 1.0), fsmall = fma(-b_last, yn, 1.0);
      const bool wide = (fbig < 0x1p30) && (fsmall > 0x1p-30);  // 16 factors per accumulator between renorms
      // lanes whose b_j is ~0 would lose the low bits of b_j*y in 1 - b_j*y: carry them along
      const bool tiny = __ballot(act && fabs(b * yn) < 0.015625) != 0ull;
      PLA_PHASE(10);
      ProdAcc acc, acc2;
      acc.init();
      acc2.init();
      const double nb = -b;
      double corr = 0.0;
      int i = 0;
      if (wide && !tiny) {
        // factors within 2^+-15: four trips (4 quad factors per accumulator each) fit between renorms
        const bool narrow = (fbig < 0x1p15) && (fsmall > 0x1p-15);
        for (; i < n32; i += 32) {
#pragma unroll
          for (int u = 0; u < 32; u += 8) {
            const double2 a12 = *reinterpret_cast<const double2*>(yp + i + u);      // e1, e2 of one quad
            const double2 a34 = *reinterpret_cast<const double2*>(yp + i + u + 2);  // e3, e4
            const double2 b12 = *reinterpret_cast<const double2*>(yp + i + u + 4);  // the next quad
            const double2 b34 = *reinterpret_cast<const double2*>(yp + i + u + 6);
            acc.mul(fma(nb, fma(nb, fma(nb, fma(nb, a34.y, a34.x), a12.y), a12.x), 1.0));
            acc2.mul(fma(nb, fma(nb, fma(nb, fma(nb, b34.y, b34.x), b12.y), b12.x), 1.0));
          }
          if (!narrow || (i & 96) == 96) {
            acc.renorm();
            acc2.renorm();
          }
        }
        acc.renorm();
        acc2.renorm();
      } else {
        for (; i < n; ++i) {
          const double yi = y[i];
          const double f = fma(nb, yi, 1.0);
          corr += fma(nb, yi, 1.0 - f) * __builtin_amdgcn_rcp(f);  // rounding error of f, relative
          acc.mul(f);
          acc.renorm();
        }
      }
      PLA_PHASE(11);
      acc.m *= acc2.m;
      acc.e += acc2.e;
      const double rn = recip_fast(nn);
      const double kj = ((log_tab(acc.m, tb.lt) + (double)acc.e * kLn2) + corr) * rn;   // psis.py:190
      const double ls = nn * (log_tab(act ? -div_fast(b, kj) : 1.0, tb.lt) - kj - 1.0);             // psis.py:191
      const double lmax = wave_all<R_MAX>(act ? ls : -INF);
      // NaN anywhere, or max = +-inf: every weight is NaN in the reference -> nothing is kept
      const bool anynan = (__ballot(act && (ls != ls)) != 0ull) || !(fabs(lmax) < INF);
      double w = act ? exp_neg(ls - lmax, tb.tab) : 0.0;                          // psis.py:192
      const double se = wave_all<R_SUM>(w);
      // the weights stay unnormalised (b_post is a ratio): w/se >= 10 eps  <=>  w >= 10 eps se
      if (anynan) w = qnan();
      const bool keep = act && (w >= (10.0 * kEps) * se);                         // psis.py:194-197
      double sw, bw;
      wave_all2<R_SUM>(keep ? w : 0.0, keep ? b * w : 0.0, sw, bw);
      const double b_post = (sw > 0.0) ? div_fast(bw, sw) : 0.0;                  // psis.py:198,201
      // psis.py:203: mean_i log1p(-b_post*y_i) as the log of per-lane products
      PLA_PHASE(12);
      double pr;
      {  // y is zero past n: no bounds
        const double f0 = fma(-b_post, y[lane], 1.0), f1 = fma(-b_post, y[lane + kWave], 1.0);
        const double f2 = fma(-b_post, y[lane + 2 * kWave], 1.0);
        pr = f0 * f1 * f2;
        if (!three) pr *= fma(-b_post, y[lane + 3 * kWave], 1.0);
        if constexpr (kBigTail) {
          if (eight) {
            double pr2 = 1.0;
#pragma unroll
            for (int u = 4; u < 8; ++u) pr2 *= fma(-b_post, y[lane + u * kWave], 1.0);
            pr *= pr2;
          }
        }
      }
      const double k_post = wave_all<R_SUM>(log_tab(pr, tb.lt)) * rn;
      const double sigma = -k_post / b_post;                                      // psis.py:205
      khat = (nn * k_post + 5.0) / (nn + 10.0);                                   // psis.py:206
      PLA_PHASE(13);
      if (isfinite(khat)) {
        const double* inv_e = sb + kSa + kWave;
        double* wtail = sb + kSa + kWave;  // weights mode: overwrites inv_e (not needed there)
        smoothed = true;
        const double rk = 1.0 / khat;
        const bool ktiny = fabs(khat) < kEps;
        if (sigma > 0.0 && !ktiny && n == M) {
          // the usual case, straight-line: w_j = sigma/k (e^{z_j} - 1) + e_cut with z_j = -k log1p(-p_j) from
          // the host table (psis.py:153,218-221).  e^z - 1 by subtraction is accurate to 1e-16 ABSOLUTE,
          // which is all the sum w_j + e_cut can see.
          const double coef = sigma * rk, off = e_cut - coef;
          const auto smooth_pass = [&](auto UC) {
            constexpr int U = decltype(UC)::value;
            double ez[U], yv[U], iv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int j = lane + kWave * u;
              const double z = fmin(-khat * l1tab[j < n ? j : 0], 700.0);
              ez[u] = exp_tab(z, tb.tab);
              yv[u] = y[j];
              iv[u] = inv_e[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int j = lane + kWave * u;
              double wj = fma(ez[u], coef, off);  // exp(log(q + e_cut)), psis.py:155
              wj = fmin(wj, 1.0);                 // psis.py:157
              const double ej = yv[u] + e_cut;
              acc_t += (j < n) ? wj - ej : 0.0;
              acc_r += (j < n) ? wj * iv[u] : 0.0;
              if constexpr (LW) wtail[j] = wj;  // (clipped) smoothed weight of tail element j
            }
          };
          if (three) smooth_pass(std::integral_constant<int, 3>{});
          else if (!eight) smooth_pass(std::integral_constant<int, 4>{});
          else if constexpr (kBigTail) smooth_pass(std::integral_constant<int, 8>{});
        } else {
          for (int j = lane; j < n; j += kWave) {
            // log1p(-p_j), p_j = (j + 0.5)/n (psis.py:153): host table when n == M
            const double l1 = (n == M) ? l1tab[j] : log_fast(1.0 - ((double)j + 0.5) * rn);
            double q;
            if (sigma <= 0.0) {
              q = qnan();                                                           // psis.py:214-215
            } else {
              q = ktiny ? -l1 : expm1_tab(-khat * l1, tb.tab) * rk;                 // psis.py:218-221
              q *= sigma;
            }
            double wj = q + e_cut;   // exp(log(q + e_cut)), psis.py:155
            if (wj > 1.0) wj = 1.0;  // psis.py:157
            const double ej = y[j] + e_cut;
            acc_t += wj - ej;
            acc_r = fma(wj, inv_e[j], acc_r);
            if constexpr (LW) wtail[j] = wj;
          }
        }
      }
    }
    // total = sum_nontail e^x + sum_tail w' = (s1 - sum_tail e) + sum_tail w'
    PLA_PHASE(14);
    // The tail's exponentials are subtracted from a sum that contains them (acc_t = sum w' - sum e): when
    // the smoothed tail is far lighter than the raw one (a draw tens of nats above the rest) that
    // cancels catastrophically, and such rows are left to the general kernel, which sums like the reference.
    double s1_all, at_all, s2_all = 0.0, ar_all = 0.0;
    if constexpr (LW) wave_all2<R_SUM>(s1, acc_t, s1_all, at_all);  // (acc_t is 0 in every lane when nothing was smoothed)
    else wave_all4<R_SUM>(s1, acc_t, s2, acc_r, s1_all, at_all, s2_all, ar_all);
    const double total = s1_all + at_all;
    if (!(total > kCancelGuard * s1_all)) slow = true;
    if constexpr (LW) {
      // ---- weights mode: lw_s = x_s - log(total) for every draw, the smoothed tail at its positions ----
      const double L = log_tab(total, tb.lt);  // psis.py:158 (_logsumexp of the shifted, smoothed row)
      if (!(total > 1e-280) || !isfinite(L)) {
        slow = true;
      } else {
        if constexpr (DEFER) {
          loo = L;
          lppd = smoothed ? (double)n : 0.0;
        } else {
          T* orow = reinterpret_cast<T*>(P.lw_out) + r * (int64_t)S;
          const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(orow, 0, S * (int)sizeof(T), 0x00020000);
          lw_store_chunk<T, VEC, false>(v, ro, ro, lane, qfull, m, L);
          if (smoothed) lw_patch_tail<T>(sm, tb, orow, lane, n, L);
        }
      }
    } else {
      s2 = s2_all;
      // loo_i = -m - L + log(tail_ratio),  L = log(total) (psis.py:158; loo.py:289,319-324): one log
      double tail_ratio = (double)S;
      if (smoothed) tail_ratio = (double)(S - n) + ar_all;
      // the two logs of the row in one call: lane 1 takes s2, every other lane the weight ratio
      const double lg = log_tab(lane == 1 ? s2 : div_fast(tail_ratio, total), tb.lt);
      loo = lane_value(lg, 0) - m;
      lppd = ((lane_value(lg, 1) + lppd_shift) - R) + ((-mn) - logS);       // loo.py:329-337
      if ((!(total > 1e-280) || !isfinite(loo) || !isfinite(lppd)) && !(dbgs & 63)) slow = true;
    }
  }
}

// ---- split pass, everything after the sweep (production path of the one-chunk LOO kernel) --------------------------------
// Same selection as wave_back() -- 512-bin histogram of the candidate list, suffix scan, candidates at / above the boundary
// bin grouped by bin -- but arranged for LATENCY, because that is what this phase costs: measured on C3, the kernel streams at
// 6.4 TB/s up to the end of the sweep and the ~750 instructions after it used to add a quarter to its run time (profiles/
// r02_phase_ablation_1m.txt): chains of dependent LDS round trips at two waves per SIMD.  Here every stage issues all its
// LDS operations at once: the (up to 512) candidates are read into 8 registers per lane ONCE and feed the histogram atomics,
// the boundary test and the scatter; the two wave-wide sums are reduced under the first LDS round trip; the hand-over's
// exponentials run as `ws_stride / 64` independent chains.  Lists longer than 512 take a second, looped batch.
// Hands over to fit_rows_kernel (pla_fit.h): y = e^x - e^xcut of the tail in bin-grouped descending order + 6 scalars.
// Where the split selection finds its candidates and its scratch: the sweep's LDS list (wave kernels), or a list in global
// memory read in place (pla_col.h: no LDS copy, so that kernel keeps three times the waves per CU).
template <class SM>
struct CandInLds {
  SM& sm;
  __device__ __forceinline__ double at(unsigned c) const { return sm.cand[c]; }
  __device__ __forceinline__ unsigned* dump_bin(int lane) const { return reinterpret_cast<unsigned*>(&sm.cand[SM::Caps::kCand]) + lane; }
  __device__ __forceinline__ double* dump_slot(int lane) const { return &sm.cand[SM::Caps::kCand + kWave + lane]; }
};

// hand-over stores.  SYNC (streamed pass): agent-scope (sc1, write-through) stores, each 128-byte line written whole by one
// instruction of this wave -- the form the fit kernel running beside this one may read behind the chunk's flag with sc1 loads
// (MI355X_MICROARCH.md, inter-workgroup visibility)
template <bool SYNC>
__device__ __forceinline__ void ws_store(double* p, double v) {
  if constexpr (SYNC) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
// the observation's scalars: lanes 0 .. sstride-1 write one entry each (one instruction; a whole line when sstride = 16)
template <bool SYNC>
__device__ __forceinline__ void ws_store_scalars(const FastParams& F, const int64_t r, const int lane, const double m, const double mn,
                                                 const double s1, const double s2, const double e_cut, const double n) {
  const int ss = F.ws_sstride;
  if (lane < ss) {
    const double v = lane == 0 ? m : (lane == 1 ? mn : (lane == 2 ? s1 : (lane == 3 ? s2 : (lane == 4 ? e_cut : (lane == 5 ? n : 0.0)))));
    ws_store<SYNC>(F.ws_s + r * ss + lane, v);
  }
}

template <typename SM, typename TB, int HU = 4, class SRC = CandInLds<SM>, bool SYNC = false>  // HU: 64-value blocks of the hand-over (ws_stride <= 64 HU)
__device__ __forceinline__ void wave_select_split(const FastParams& F, SM& sm, const TB& tb, const int64_t r, const int lane_id,
                                                  const int M, const double m, const double mn, const double s1, const double s2,
                                                  const unsigned ncand, const int k1, const int sh, const double magic,
                                                  const double c256, bool& slow, const SRC& cs, const int dbgs = 0) {
  constexpr int kSa = SM::Caps::kSa;
  constexpr int U = 8;  // candidates per lane held in registers
  // (opaque: the addresses this phase derives from the lane number must not be computed -- and kept alive -- across the sweep)
  int lane = lane_id;
  asm volatile("" : "+v"(lane));
  const auto key_of = [&](double xx) { return __double2loint(fma(xx, c256, magic)); };
  // Lanes with nothing to count or to place still execute the stage's LDS operations (straight-line code, every round trip in
  // flight together) on a dump bin / dump slot of their OWN -- 64 lanes hammering one LDS address cost more than the
  // branches they save.  Both live in the tail of the candidate array, which the sweep no longer needs: its 64 overflow
  // slots (as 32-bit bins) and its per-lane dump slots.
  unsigned* const dump_bin = cs.dump_bin(lane);
  double* const dump_slot = cs.dump_slot(lane);
  PLA_PHASE(4);
  // ---- candidates -> registers; histogram atomics (no return value: fire and forget) -------------------------------
  double xs[U];
  int bx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned c = lane + kWave * u;
    xs[u] = cs.at(c < ncand ? c : 0);
  }
  double s1_all, s2_all;
  wave_all2<R_SUM>(s1, s2, s1_all, s2_all);  // (independent of the list: runs while the reads are in flight)
  const auto ablate_exit = [&](int bit) {  // profiling builds: stop here, leave a tail of length 0 (not fitted)
    if (!(dbgs & bit)) return false;
    ws_store_scalars<SYNC>(F, r, lane, m, mn, s1_all, s2_all, 0.5, 0.0);
    return true;
  };
  if (ablate_exit(32)) return;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned c = lane + kWave * u;
    bx[u] = c < ncand ? (key_of(xs[u]) - k1) >> sh : kWaveBins;  // (kWaveBins: past the end of the list)
    atomicAdd(bx[u] < kWaveBins ? &sm.hist[bx[u]] : dump_bin, 1u);
  }
  for (unsigned c = lane + kWave * U; c < ncand; c += kWave) atomicAdd(&sm.hist[(key_of(cs.at(c)) - k1) >> sh], 1u);
  wave_sync();
  PLA_PHASE(5);
  // ---- suffix scan over the bins (8 per lane): boundary bin of rank M ----------------------------------------------
  int bstar = 0, C1 = 0, nbnd = 0;
  {
    unsigned c[8];
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 h = *reinterpret_cast<const uint4*>(&sm.hist[8 * lane + 4 * i]);
      c[4 * i] = h.x; c[4 * i + 1] = h.y; c[4 * i + 2] = h.z; c[4 * i + 3] = h.w;
      tot += h.x + h.y + h.z + h.w;
    }
    unsigned pre = tot;
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x111, 0xF, 0xF, true);   // row_shr:1
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x112, 0xF, 0xF, true);   // row_shr:2
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)tot, 0x113, 0xF, 0xF, true);   // row_shr:3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x114, 0xF, 0xE, false);  // row_shr:4, banks 1-3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x118, 0xF, 0xC, false);  // row_shr:8, banks 2-3
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x142, 0xA, 0xF, false);  // row_bcast:15
    pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x143, 0xC, 0xF, false);  // row_bcast:31
    const unsigned all = (unsigned)__builtin_amdgcn_readlane((int)pre, kWave - 1);
    unsigned a = all - pre;  // candidates in bins owned by higher lanes
    int fb = -1, fc = 0, fn = 0;
    unsigned st[8];
#pragma unroll
    for (int i = 7; i >= 0; --i) {
      st[i] = a;
      if ((unsigned)M >= a && (unsigned)M < a + c[i]) {
        fb = 8 * lane + i;
        fc = (int)(a + c[i]);
        fn = (int)c[i];
      }
      a += c[i];
    }
    *reinterpret_cast<uint4*>(&sm.start[8 * lane]) =
        make_uint4(st[0] | (st[1] << 16), st[2] | (st[3] << 16), st[4] | (st[5] << 16), st[6] | (st[7] << 16));
    const unsigned long long who = __ballot(fb >= 0);
    const int src = __ffsll((long long)who) - 1;
    bstar = __builtin_amdgcn_readlane(fb, src);
    C1 = __builtin_amdgcn_readlane(fc, src);
    nbnd = __builtin_amdgcn_readlane(fn, src);
  }
  wave_sync();
  if (C1 > kSa || nbnd > kWave) {  // (more than 64 draws share the cutoff's bin: heavy ties)
    slow = true;
    return;
  }
  if (ablate_exit(64)) return;
  PLA_PHASE(6);
  // ---- candidates at / above the boundary bin -> sa, grouped by bin (descending bins): all the slot requests at once -----
  // (the candidates are read from the list a second time rather than held in registers across the scan: that stretch is where
  // the kernel's register count peaks, and sixteen registers less there let the fit kernel of the previous block of
  // observations run beside this kernel -- pla_capi.hip, pipelined pass)
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned c = lane + kWave * u;
    xs[u] = cs.at(c < ncand ? c : 0);
  }
  unsigned old[U], st[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool take = bx[u] >= bstar && bx[u] < kWaveBins;
    old[u] = atomicSub(take ? &sm.hist[bx[u]] : dump_bin, 1u);
    st[u] = sm.start[take ? bx[u] : 0];
    bx[u] = take ? 1 : 0;
  }
#pragma unroll
  for (int u = 0; u < U; ++u) *(bx[u] ? &sm.sa[st[u] + old[u] - 1u] : dump_slot) = xs[u];
  for (unsigned c = lane + kWave * U; c < ncand; c += kWave) {  // (lists beyond 512 entries)
    const double x = cs.at(c);
    const int b = (key_of(x) - k1) >> sh;
    if (b >= bstar) sm.sa[sm.start[b] + (atomicSub(&sm.hist[b], 1u) - 1u)] = x;
  }
  wave_sync();
  if (ablate_exit(128)) return;
  PLA_PHASE(8);
  // ---- the boundary bin: it holds the cutoff x_(S-M) itself (psis.py:135-136); its members strictly above the cutoff are
  //      the last of the tail (psis.py:139: ties at the cutoff leave the tail) ------------------------------------------
  const int na = C1 - nbnd;  // candidates in the bins above: all in the tail
  const double xb = sm.sa[na + (lane < nbnd ? lane : 0)];
  int gt = 0, ge = 0;
  for (int j = 0; j < nbnd; ++j) {
    const double xj = lane_value(xb, j);
    gt += (xj > xb) ? 1 : 0;
    ge += (xj >= xb) ? 1 : 0;
  }
  const int want = M - na;  // the cutoff is the (M - na)-th largest (0-based) of the boundary bin: gt <= want < ge
  const unsigned long long isc = __ballot(lane < nbnd && gt <= want && want < ge);
  const int src = __ffsll((long long)isc) - 1;
  const double xcut = lane_value(xb, src);
  const int n = na + __builtin_amdgcn_readlane(gt, src);  // draws strictly above the cutoff
  // ---- hand-over: the tail's shifted log ratios x (psis.py:139) in the candidates' order, the cutoff itself from n up to the row
  // stride (it sorts behind every tail value and its y is exactly 0), and the cutoff; the exponentials y = e^x - e^xcut (psis.py:147) are the fit kernel's, which has lanes to spare for them ----
  // The boundary bin's members above the cutoff close ranks behind the higher bins IN LDS (every lane holds its xb by now), so
  // that the row goes out as whole 512-byte stores, nothing scattered behind them.
  const int stride = F.ws_stride;
  double* wy = F.ws_y + r * (int64_t)stride;
  if (n > 4) {
    const bool mine = lane < nbnd && xb > xcut;
    const unsigned long long mm = __ballot(mine);
    const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mm, 0u));
    *(mine ? &sm.sa[na + (int)pos] : dump_slot) = xb;
    wave_sync();
    double xh[HU];
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int j = lane + kWave * u;
      xh[u] = sm.sa[j < n ? j : 0];
    }
    int nblk = stride;  // (opaque per row: else the HU guards are evaluated above the row loop and held in scalar registers the
    asm volatile("" : "+s"(nblk));  // long-row kernels do not have)
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int j = lane + kWave * u;
      if (kWave * u < nblk && !(dbgs & 256)) ws_store<SYNC>(wy + j, j < n ? xh[u] : xcut);  // (wave-uniform guard)
    }
  }
  ws_store_scalars<SYNC>(F, r, lane, m, mn, s1_all, s2_all, xcut, (double)n);
}

// LW = false: LOO mode (input = log-likelihood, raw = -ll; outputs k-hat, loo_i, lppd_i)
// LW = true:  weights mode (input = log ratios, raw = input; outputs k-hat and the normalised smoothed
//             log-weights, psis.py:78-111): the row stays in its registers until the weights are stored
template <typename T, int VEC, bool LW, typename SM, typename TB, bool SPLIT = false, bool SYNC = false>
__device__ PLA_ROW_INLINE void wave_loo_row(const RowsParams& P, const FastParams& F, SM& sm, const TB& tb, const int64_t r,
                                            T (&v)[kWaveSlots], const T* rp_next) {
  constexpr int EPT = kWaveSlots;
  constexpr int NQ = EPT / VEC;
  constexpr int kCand = SM::Caps::kCand, kSa = SM::Caps::kSa;  // LDS capacities of this instantiation
  // (opaque per row: lane-derived masks and addresses of the later phases are then computed where they are used instead of
  // being hoisted above the row loop, where they would sit on top of the 128 row registers)
  int lane = wave_lane();
  asm volatile("" : "+v"(lane));
  // parameters arrive by reference (memory): read each once into scalar registers
  const int S = __builtin_amdgcn_readfirstlane(P.n_draws);
  const int M = __builtin_amdgcn_readfirstlane(P.tail_count);
  const int gsz = __builtin_amdgcn_readfirstlane(F.gsz);
  const int kq = __builtin_amdgcn_readfirstlane(F.kq);
#if PLA_WAVE_ABLATE
  const int dbgs = __builtin_amdgcn_readfirstlane(F.debug_skip);
#else
  constexpr int dbgs = 0;
#endif
  const int mestM = __builtin_amdgcn_readfirstlane(F.mest_M);
  const double logS = uniform_d(F.log_S);
  const double INF = pinf();
  const int nvec = S / VEC;                       // 16-byte vectors per row (S % VEC == 0)
  const int qfull = nvec / kWave;                 // q < qfull: every lane valid
  const int qrem = nvec - qfull * kWave;          // q == qfull: lanes < qrem valid

#if defined(PLA_PHASE_CLOCK)
  unsigned long long clk_a = __builtin_amdgcn_s_memtime();
#define PLA_CLK(k)                                                   \
  do {                                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    if (lane == 0) sm.clk[k] += now_ - clk_a;                        \
    clk_a = now_;                                                    \
  } while (0)
#else
#define PLA_CLK(k) ((void)0)
#endif
  // ---- finish the load issued by the previous iteration (or the prologue): pad fix-up -------------
  {
    // slots past the row: copy this lane's first vector (harmless for max / min / threshold)
    // (from the last vector down, leaving at the first complete one: a chain of wave-uniform early
    // exits instead of NQ predicated selects -- usually a single vector is partial)
    pad_tail<T, VEC, NQ - 1, false>(v, qfull, qrem, (T)0);
  }
  // ---- 1. row statistics, in the input precision (exact; duplicates of valid draws are harmless) ---
  // raw = -ll:  max raw = max(-v),  min raw = -max(v);  gs = max raw over the first `gsz` slots this lane VISITS (a sample
  // spread over the row, see bitrev_order)
  double mx, mn, gs;
  row_stats<T, VEC, LW>(v, gsz, __builtin_amdgcn_readfirstlane(F.sample_bits), mx, mn, gs);
  PLA_CLK(0);  // (statistics: includes the wait for the row's loads)
  double m, nmn, ngs, unused_;
  wave_all4<R_MAX>(mx, -mn, -gs, -gs, m, nmn, ngs, unused_);  // min = -max(-.): row max, row min, smallest group maximum
  mn = -nmn;
  const double R = m - mn;
  // Speculative candidate threshold: a value with at least `kq` of the 64 per-lane group maxima
  // below it, found by bisection on ballots.  For exchangeable draws a fraction ~(kq/64)^(1/gsz) of
  // the row lies below it, i.e. a few hundred draws lie above (the launcher picks gsz and kq so that
  // this is ~2.2(M+1)); rows where the guess is off are recomputed by the general kernel.
  double t1;
  {
    // bisection in fixed point: the group maxima as integers in [0, 2^20] (one VALU compare per step, the
    // interval lives in scalar registers)
    const double lo0 = -ngs, spread = m - lo0;
    const int ki = (int)((gs - lo0) * (1048576.0 * recip_fast(spread)));  // spread = 0 (constant sample): t1 = 0 -> general kernel
    int lo_i = 0, hi_i = 1 << 20;
#pragma unroll
    for (int it = 0; it < PLA_BISECT_ITERS; ++it) {  // 2^-iters of the spread of the group maxima: a handful of candidates
      const int mid = (lo_i + hi_i) >> 1;
      const int below = __popcll(__ballot(ki < mid));
      if (below >= kq) hi_i = mid; else lo_i = mid;
    }
    t1 = fma((double)hi_i, spread * (1.0 / 1048576.0), lo0) - m;
  }
  // pads: from here on the slots past the row hold ll = -mn (raw = mn, x = -R: the smallest value of the row) instead of copies
  // of the lane's first vector -- never at or above a threshold, so the counts below see real draws only
  // (LOO mode; the weights kernel does it right before the sweep and counts valid slots only)
  if constexpr (!LW) pad_tail<T, VEC, NQ - 1, true>(v, qfull, qrem, (T)(-mn));
  bool thr_ok = true;
  if (m - mn < kWaveMaxRange) {  // (rows with non-finite draws or too wide a range go to the general kernel anyway)
    double t_raw = t1 + m;
    thr_ok = wave_threshold_check<T, VEC, LW>(v, F, mn, m, t_raw);
    t1 = t_raw - m;
  }
  // +-inf in the row makes R inf/NaN; a NaN draw is ignored by v_max here, poisons s1 in the sweep
  // and is caught by the finiteness test at the end: both land on the general kernel
  PLA_PHASE(1);
  PLA_CLK(1);  // (reductions, threshold, exact count)
  bool slow = !(R < kWaveMaxRange) || !(t1 < 0.0) || !thr_ok;
  // constants every later phase uses, pinned in registers by hand (MachineLICM is off for this file:
  // the compiler would otherwise re-materialise them inside every loop)
  double magic = kMagic, c256 = kC256;
  asm volatile("" : "+v"(magic));
  asm volatile("" : "+s"(c256));
  const auto key_of = [&](double xx) { return __double2loint(fma(xx, c256, magic)); };  // = key256, hoisted
  const int k1 = key_of(t1);                 // histogram origin
  const int kpad = key_of(-R);               // key of the pad value (smallest x of the row)
  if (kpad >= k1) slow = true;               // pads would be counted as candidates
  double khat = INF, loo = 0.0, lppd = 0.0;
  bool streamed = false;  // next row's loads already issued (inside the sweep)
#if PLA_WAVE_ABLATE
  int why_cand = 0;
#endif
  // LOO mode sweeps EVERY row, also the ones already lost to the general kernel (their sums and candidates are garbage and
  // are not looked at): the sweep is where the next row's loads are issued, and a second place of issue for the rows that
  // skip it made the register allocator merge two copies of the row at the end of every row -- a vmcnt(0) wait and 116
  // register moves per row in the generated code.  (Weights mode issues them in one place anyway, after its output pass.)
  if (dbgs & 16) {
    loo = t1 + R;  // ablation: statistics and threshold only
  } else if (!LW || !slow) {
    // bins over the candidates: (k - k1) >> sh  in [0, 511] for k in [k1, 0]
    const int span = -k1;
    const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
    wave_sync();  // previous row is done with the LDS scratch
    {
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int i = 0; i < kWaveBins / (4 * kWave); ++i) *reinterpret_cast<uint4*>(&sm.hist[4 * (lane + kWave * i)]) = z4;
    }
    if constexpr (LW) pad_tail<T, VEC, NQ - 1, true>(v, qfull, qrem, (T)mn);  // pads: raw = mn (x = -R)
    wave_sync();
    // ---- 2. sweep: e^x and e^-x of every draw from one range reduction + histogram of candidates ----
    // (e^(ll - max ll) = e^-R * e^-x; the row constant e^-R is applied to the sum, in log space)
    PLA_PHASE(2);
    double s1 = 0.0, s2 = 0.0;
    // the constants of the sweep live in SGPRs for its whole length (with MachineLICM off the compiler
    // would re-materialise each of them with s_mov per use)
    double nl256 = -kLn2_256, c6 = 1.66666666666666666667e-01;
    int c4096 = 4096, cm4096 = -4096;
    asm volatile("" : "+s"(nl256), "+s"(c6), "+s"(c4096), "+s"(cm4096));
    int four = 4;
    asm volatile("" : "+v"(four));
    // the next row (a zero-length range when there is none: the loads then return zeros and touch nothing)
    // (weights mode, rounds 1-3, streamed the SAME row in again -- three row passes; PLA_LW_RESTREAM = 1 brings that back)
    const T* rp_stream = LW ? reinterpret_cast<const T*>(P.in) + PLA_ROW_OFFSET(P, r) : rp_next;
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(rp_stream ? rp_stream : (const T*)P.in), 0, rp_stream ? S * (int)sizeof(T) : 0, 0x00020000);
    streamed = !LW;  // weights mode: the next row is requested after the weights have been stored
    const char* tabc = reinterpret_cast<const char*>(tb.tab);
    // The candidate list is addressed with raw LDS byte addresses so that this wave's scratch base
    // rides in the scalar append offset / the precomputed dump address instead of costing a vector
    // add per draw.
    const unsigned cand0 = lds_addr(sm.cand);
    const unsigned dump8 = cand0 + (unsigned)(kCand + kWave + lane) * 8u;  // this lane's dump slot
    const unsigned lim8 = cand0 + 8u * kCand;
    unsigned next8 = cand0;                               // address of the next append (unclamped)
    unsigned base8 = cand0;                               // min(next8, lim8)
    // Software pipeline, kPF draws deep: stage A of draw i+kPF (shift, range reduction, table read,
    // candidate append) is issued before stage B of draw i (polynomial, accumulate), so the LDS
    // latency of the table read is covered by the arithmetic of the draws in between.
    constexpr int kPF = PLA_SWEEP_DEPTH;
    int sbase = 0;  // scalar part of the next row's load offsets
    double px[kPF], pt[kPF];
    int4 ptt[kPF];
#pragma unroll
    for (int i = 0; i < EPT + kPF; ++i) {
      if (i >= kPF) {  // stage B of draw i - kPF: 11 VALU
        const int sl = (i - kPF) % kPF;
        const double x = px[sl], t = pt[sl];
        const int k = __double2loint(t);
        const double rr = fma(t - magic, nl256, x);        // |rr| <= ln2/512
        const double r2 = rr * rr;
        const double E = fma(r2, 0.5, 1.0);                // cosh rr to 1.5e-13 (r^4/24 dropped)
        const double O = fma(c6, r2, 1.0);                 // sinh rr / rr
        if (!(dbgs & 1)) {
          s1 = fma(__hiloint2double(mad_i24(k, c4096, ptt[sl].y), ptt[sl].x), fma(rr, O, E), s1);
          s2 = fma(__hiloint2double(mad_i24(k, cm4096, ptt[sl].w), ptt[sl].z), fma(-rr, O, E), s2);
        }
        // pin the running sums: otherwise the accumulation chain is sunk to the end of the block and
        // its inputs (table entries, reduced arguments) spill
        if ((i & 3) == 3) asm volatile("" : "+v"(s1), "+v"(s2));
      }
      if (i < EPT) {  // stage A of draw i: 8 VALU
        const int sl = i % kPF;
        const double x = LW ? (double)v[i] - m : (-(double)v[i]) - m;  // psis.py:134
        const double t = fma(x, c256, magic);
        const int k = __double2loint(t);       // round(x * 256/ln2): low mantissa bits of t
        px[sl] = x;
        pt[sl] = t;
        ptt[sl] = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));  // 16 * (k & 255)
        // candidates (x >= t1, ~1 draw in 7) are appended to the LDS list.  One wave owns the list, so
        // the slot is a running scalar count + the lane's rank among this draw's candidates: no
        // atomic, no LDS round trip.  Everybody else writes to the lane's private dump slot, which
        // keeps the sweep free of branches; an overflowing list spills into the 64 slots behind it.
        const bool cand = x >= t1;
        const unsigned long long cm = __ballot(cand);
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
        unsigned pos8 = (rank << 3) + base8;
        asm("" : "+v"(pos8));  // computed by every lane: a select below, not a divergent region
        lds_store(cand ? pos8 : dump8, x);
        if constexpr (LW)  // draw index of the candidate: VEC * (lane + 64 q) + e for slot i = q VEC + e
          sm.ids[cand ? ((pos8 - cand0) >> 3) : (unsigned)(kCand + kWave + lane)] =
              (unsigned short)(VEC * lane + (VEC * kWave * (i / VEC) + i % VEC));
        {  // next8 += 8 * popcount(cm) in two scalar ops (the compiler would re-associate it into four)
          const unsigned pc = (unsigned)__popcll(cm);
          asm("s_lshl3_add_u32 %0, %1, %0" : "+s"(next8) : "s"(pc) : "scc");
        }
        base8 = next8 < lim8 ? next8 : lim8;
        // This was the last read of slot i.  Once a whole 16-byte vector has been consumed, the next
        // row's vector is streamed into the same registers: the loads of row r+1 trickle out during
        // the sweep of row r and have the whole selection / fit / smoothing phase to arrive, without a
        // single extra register and without a burst that would stall every wave of the CU at once.
        if ((i % VEC) == VEC - 1) {
          constexpr int kPerBase = 4096 / (kWave * 16);  // vectors per 4 KB of row
          const int q = i / VEC;
          if (q % kPerBase == 0) {
            sbase = q * (kWave * 16);
            asm volatile("" : "+s"(sbase));  // one scalar per four vectors (the compiler would materialise one per load)
          }
          if constexpr (!LW || PLA_LW_RESTREAM) issue_row_vector<T, VEC, true>(v, rs_next, q, sbase);
        }
      }
    }
    if constexpr (LW && !PLA_LW_RESTREAM) {
      // weights mode keeps the row where it is: two row passes (one read, one write).  The registers are made opaque here so
      // that the output pass recomputes x = raw - m from them instead of keeping the sweep's 64 shifted values alive
#pragma unroll
      for (int i = 0; i < EPT; ++i) asm volatile("" : "+v"(v[i]));
    }
    const unsigned ncand = (next8 - cand0) >> 3;
    PLA_PHASE(3);
    PLA_CLK(2);  // (sweep)
    {  // remove the pads' contribution (same code path, so it cancels to rounding)
      const double x = -R;
      const double t = fma(x, kC256, magic);
      const int k = __double2loint(t);
      const double rr = fma(t - magic, -kLn2_256, x);
      const int4 tt = *reinterpret_cast<const int4*>(tabc + 16 * (k & 255));
      const double r2 = rr * rr;
      const double E = fma(r2, 0.5, 1.0);
      const double O = fma(1.66666666666666666667e-01, r2, 1.0);
      const double npad = (double)((NQ - qfull) * VEC - ((lane < qrem) ? VEC : 0));  // padded slots of this lane
      if (!(dbgs & 1)) {
        s1 = fma(-npad * __hiloint2double(tt.y + (k << 12), tt.x), fma(rr, O, E), s1);
        s2 = fma(-npad * __hiloint2double(tt.w - (k << 12), tt.z), fma(-rr, O, E), s2);
      }
    }
    wave_sync();
    if (slow) {
      // (nothing: the row is on its way to the general kernel)
    } else if (dbgs & 4) {
      loo = s1;
      lppd = s2;
    } else if ((int)ncand < M + 1 || ncand > (unsigned)kCand) {
      slow = true;  // the speculative threshold missed (too few / too many draws above it)
#if PLA_WAVE_ABLATE
      why_cand = (int)ncand < M + 1 ? 1 : 2;
#endif
    } else if constexpr (SPLIT && kFitSorts && !LW) {
      wave_select_split<SM, TB, 4, CandInLds<SM>, SYNC>(F, sm, tb, r, lane, M, m, mn, s1, s2, ncand, k1, sh, magic, c256, slow,
                                                        CandInLds<SM>{sm}, dbgs);
    } else {
      static_assert(!SPLIT, "split pass: LOO mode only, through wave_select_split");
      wave_back<T, VEC, LW, SM, TB>(P, sm, tb, r, v, lane, S, M, mestM, logS, dbgs, m, mn, R, 0.0, s1, s2, ncand, k1, sh, magic, c256,
                                    qfull, qrem, slow, khat, loo, lppd);
    }
  }
  // the next row starts streaming into the (now dead) row registers while the outputs are stored and
  // the other wave of this SIMD computes
  PLA_PHASE(15);
  PLA_CLK(3);  // (selection and hand-over)
#if defined(PLA_PHASE_CLOCK)
  if (lane == 0) sm.clk[4] += 1ull;
#endif
  // (tail length -1: tells the fit kernel that this observation is on the list for the general kernel)
  if constexpr (SPLIT) {
    if (slow) ws_store_scalars<SYNC>(F, r, lane, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0);
  }
  if constexpr (LW || PLA_WAVE_ABLATE) {
    if (!streamed && rp_next) issue_row_loads<T, VEC, (LW && PLA_LW_RESTREAM) ? 0 : PLA_LOAD_AUX>(v, rp_next, S);  // (weights mode: every row)
  }
  if (lane == 0) {
    if (slow) {
      const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
      F.slow_list[idx] = (unsigned)r + F.slow_base;
#if PLA_WAVE_ABLATE
      {  // why the row left the fast path (profiling builds: PLA_PRINT_REASONS)
        int why = 7;  // selection / outputs
        if (!(R < kWaveMaxRange)) why = 0;
        else if (!thr_ok) why = 1;
        else if (!(t1 < 0.0)) why = 2;
        else if (kpad >= k1) why = 3;
        else if (why_cand) why = 3 + why_cand;  // 4: too few candidates, 5: too many
        atomicAdd(&F.counters[8 + why], 1ull);
      }
#endif
    } else if constexpr (!SPLIT) {
      if (P.diag) P.diag[r] = khat;
      if constexpr (!LW) {
        if (P.loo_i) P.loo_i[r] = P.scale_value * loo;
        if (P.lppd_i) P.lppd_i[r] = lppd;
      }
    }
  }
}

// The per-row body is deliberately NOT inlined into the row loop: inlined, LLVM hoists every
// loop-invariant constant, mask and offset of the later phases above the loop, where they sit on
// top of the 128 row registers and spill.
template <typename T, int VEC, bool LW, class CAP, bool SPLIT = false, bool SYNC = false>
__global__ __launch_bounds__(kWave * CAP::kWaves, PLA_MIN_WAVES_PER_SIMD) void wave_loo_kernel(RowsParams P, FastParams F) {
  using SM = std::conditional_t<LW, WaveSmemLWT<CAP>, WaveSmemT<CAP>>;
  // (the selection half of the split pass needs the exponential table only: 6.5 KB of LDS less per workgroup, which is what lets
  // a four-wave workgroup of the fit kernel sit beside two of these on a CU)
  constexpr bool kTabOnly = SPLIT && kFitSorts && !LW;
  using TB = std::conditional_t<kTabOnly, WaveTabOnly, WaveTablesT<CAP>>;
  constexpr int kWavesPerBlock = CAP::kWaves;
  __shared__ __attribute__((aligned(16))) SM scratch[kWavesPerBlock];
  __shared__ __attribute__((aligned(16))) TB tb;
  const int tid = threadIdx.x;
  for (int j = tid; j < kTabN; j += kWave * kWavesPerBlock) exp_table_entry(tb.tab, j);
  if constexpr (!kTabOnly) {
    for (int j = tid; j < kLogTabN; j += kWave * kWavesPerBlock) log_table_entry(tb.lt, j);
    for (int j = tid; j < P.tail_count; j += kWave * kWavesPerBlock) tb.l1[j] = F.l1_table[j];
    if (tid < kWave) tb.bg[tid] = F.b_grid[tid];
  }
#if defined(PLA_PHASE_CLOCK)
  if ((tid & (kWave - 1)) < 8) scratch[tid / kWave].clk[tid & 7] = 0ull;
#endif
  __syncthreads();  // the only workgroup barrier: from here on the waves are independent
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);  // wave-uniform, and the compiler knows it
  SM& sm = scratch[wv];
  T v[kWaveSlots];
  const T* base = reinterpret_cast<const T*>(P.in);
  if constexpr (SYNC) {
    if (F.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (F.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (F.prio == 3) __builtin_amdgcn_s_setprio(3);
  }
  const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nw = (int64_t)gridDim.x * kWavesPerBlock;
#if PLA_WAVE_ABLATE
  unsigned long long ck0, rt0;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck0), "=s"(rt0));
#endif
  // Which rows this wave takes: with a dynamic row queue (FastParams::queue) chunks of kQueueChunk consecutive rows -- the
  // chunk after the current one is known a chunk ahead, so that the last row of a chunk streams the first row of the next one
  // in behind its sweep like any other row; without one, row first + i * waves.  ONE loop and one call site for both: a
  // second copy of the row body costs the kernel fifty registers.
  const bool queued = !LW && F.queue != nullptr;
  // (the queue's state in 32-bit scalars -- rows < 2^31 with a queue -- the row loop is short of scalar registers)
  const auto dequeue = [&]() -> unsigned {
    unsigned got = 0;
    if (wave_lane() == 0) got = atomicAdd(F.queue, (unsigned)kQueueUnit);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)got);
  };
  const int64_t n = P.n_obs;
  int64_t r = w0;
  unsigned nxt = 0, chunk0 = 0;
  int left = 0;  // rows of the current unit after row r
  if (queued) {
    chunk0 = dequeue();
    nxt = dequeue();
    left = kQueueUnit - 1;
    r = (int64_t)chunk0;
  }
  if (r < n) issue_row_loads<T, VEC, (LW && PLA_LW_RESTREAM) ? 0 : PLA_LOAD_AUX>(v, base + PLA_ROW_OFFSET(P, r), P.n_draws);
  // (the unit after the next is asked for BEFORE the last row of the current unit, not behind it: a returning atomic on a
  // counter that 2048 waves share takes 1-3 us under this load, and waiting for it at the end of every unit was 0.8 % of the
  // pass at 16 rows per unit -- with smaller units, which shorten the ragged end of the launch, it was everything: 4 rows per
  // unit 8.1 ms against 6.4)
#ifndef PLA_DEQUEUE_AHEAD
#define PLA_DEQUEUE_AHEAD 1
#endif
  unsigned pend = 0;  // lane 0: the atomic's return value, in flight while the unit's last row is processed
#pragma unroll 1
  while (r < n) {
    const int64_t rn = !queued ? r + nw : (left > 0 ? r + 1 : (int64_t)nxt);
    if constexpr (PLA_DEQUEUE_AHEAD != 0 && kQueueUnit > 1) {
      if (queued && (left == 0 || r + 1 >= n)) {
        if (wave_lane() == 0) pend = atomicAdd(F.queue, (unsigned)kQueueUnit);
      }
    }
    wave_loo_row<T, VEC, LW, SM, TB, SPLIT, SYNC>(P, F, sm, tb, r, v, rn < n ? base + PLA_ROW_OFFSET(P, rn) : nullptr);
    if (queued) {
      if (left > 0 && r + 1 < n) {
        left -= 1;
      } else {
        left = kQueueUnit - 1;
        unsigned after;
        if constexpr (PLA_DEQUEUE_AHEAD != 0 && kQueueUnit > 1) after = (unsigned)__builtin_amdgcn_readfirstlane((int)pend);
        else after = dequeue();  // (its returned value is waited for: everything this wave has stored so far has drained)
        if constexpr (SYNC) {
          // streamed pass: the rows [chunk0, chunk0 + unit) are complete -- every hand-over store of theirs (sc1, whole lines) has
          // left this wave -- and count towards their chunk; the fit kernel beside this one takes a chunk once all its rows count
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (said explicitly: the compiler may know the counter to be empty and drop its own)
          if (wave_lane() == 0) {
            if constexpr (kQueueUnit == kQueueChunk) {
              __hip_atomic_store(F.done + chunk0 / kQueueChunk, (unsigned)kQueueChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              const int64_t rest = n - (int64_t)chunk0;
              (void)__hip_atomic_fetch_add(F.done + chunk0 / kQueueChunk, rest < kQueueUnit ? (unsigned)rest : (unsigned)kQueueUnit,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        chunk0 = nxt;
        nxt = after;
      }
    }
    r = rn;
  }
#if defined(PLA_PHASE_CLOCK)
  if (wave_lane() < 5) atomicAdd(&F.counters[8 + wave_lane()], sm.clk[wave_lane()]);  // (profiling slots of the counters)
#endif
#if PLA_WAVE_ABLATE
  if (blockIdx.x == 0 && tid == 0) {  // core clock against the 100 MHz real-time counter
    unsigned long long ck1, rt1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(ck1), "=s"(rt1));
    F.counters[2] = ck1 - ck0;
    F.counters[3] = rt1 - rt0;
  }
#endif
}

}  // namespace pla
