// PSIS-LOO for observations-fastest log-likelihood matrices (the layout `pl.loo(idata)` hands over, loo.py:189), second form:
// a WORKGROUP PER 16 NEIGHBOURING OBSERVATIONS, candidate lists in LDS.
//
// pla_col.h gives every observation a lane and must therefore keep the candidate lists in HBM (8 KB per observation written
// and read again: 1.5 of its 7.3 ms sweep, plus a separate selection kernel that is all latency) and re-reads a 12.8 % sample of
// the matrix from HBM for its threshold.  Here the unit is the 128-byte piece of a draw that 16 neighbouring observations share
// (tools/microbench/col_stream.hip: pieces of that size stream at the same 6.5 TB/s as the 512-byte pieces of the lane-per-
// observation sweep): one workgroup of eight waves owns the 16 observations, lane = (observation l & 15, draw l >> 4), a wave
// load reads the pieces of four consecutive draws, the eight waves cover 32 draws per step.  32 lanes work on every
// observation, so its candidates fit the CU: 16 lists of 640 entries in LDS, appended to with LDS atomics, and the selection
// of the split pass (wave_select_split, pla_wave.h) runs on them in place, two observations per wave -- nothing but the
// hand-over to fit_rows_kernel goes back to HBM.  The threshold sample (512 draws spread over the row, 16 per lane) is the
// head of the sweep itself: every wave starts its cyclic walk over the row at a place of its own.
//
//   per group of 16 observations:
//   A  sample: 16 draws per lane, spread over the row (every chain of a chain-major stack contributes), as f32 keys to LDS
//   B  threshold: the sample's ks-th largest value, by two levels of a 64-bin histogram (one wave per observation)
//   C  sweep: every draw once -- max / min, the two sums of e^x', e^-x' about the provisional shift (the sample's maximum), and
//      the draws at or above the threshold appended to the observation's list
//   D  selection on the list with the true shift (psis.py:134: x = raw - max raw in one rounding), hand-over to the fit kernel
#pragma once

#include "pla_col.h"

namespace pla {

#ifndef PLA_TILE_CAP
#define PLA_TILE_CAP 640   // candidate list capacity per observation
#endif
#ifndef PLA_TILE_RING
#define PLA_TILE_RING 16   // steps in flight per lane (= the sample's 16: C3 8.0-8.3 ms; 18: the same; 20: 8.4-8.5; 24: 8.6-8.8; 32: 8.45)
#endif
#ifndef PLA_TILE_PF
#define PLA_TILE_PF 4      // depth of the sweep's software pipeline (draws between a draw's LDS requests and their use)
#endif
#ifndef PLA_TILE_ABLATE
#define PLA_TILE_ABLATE 0  // timing experiments (tools/build_alt.sh): 1 no selection, 2 no sample / threshold, 4 sweep = loads + min only, 8 no list appends, 16 no exponentials
#endif
#ifndef PLA_TILE_WAVES
#define PLA_TILE_WAVES 8   // waves that sweep (8 of them also search the thresholds and select)
#endif
constexpr int kTileWaves = PLA_TILE_WAVES;
constexpr int kTileSelWaves = 8;
constexpr int kTileThreads = kWave * kTileWaves;
constexpr int kTileSample = 64 * kTileWaves;  // sampled draws per observation: 16 per lane
constexpr int kTileCap = PLA_TILE_CAP;
#ifndef PLA_TILE_CAP_STREAM
#define PLA_TILE_CAP_STREAM 520   // ... in the streamed pass, where the fit kernel's workgroup needs 39.5 KB of the CU's LDS
#endif
constexpr int kTileCapStream = PLA_TILE_CAP_STREAM;

struct TileParams {
  const void* in;      // element (observation i, draw s) at in[s * ld + i]
  int64_t n_obs;       // observations of this launch
  int n_draws;
  int64_t ld;          // elements between consecutive draws
  int ks;              // the threshold leaves at most ks of the 512 sampled draws at or above it
};

// LDS of one workgroup.  What only the sweep needs (the dump counters and slots of the lanes without a candidate, the per-wave
// partial sums) lies in the selection's scratch, which is idle then: with the lists at 520 entries the workgroup takes
// 119.6 KB, and a four-wave workgroup of the streamed fit kernel (39.5 KB, pla_fit.h) fits on the CU beside it.
template <typename T, int CAP>
struct TileSmem {
  static constexpr int kObs = 128 / (int)sizeof(T);  // observations per group: one 128-byte piece of a draw
  static constexpr int kCap = CAP;
  using Sel = ColSmem<CapsSmall>;
  struct Sweep {
    unsigned dumpc[kTileThreads];         // where the lanes whose draw is no candidate count (from 2^31 up: "past the end of any list")
    T dumpv[kTileThreads];                // ... and store
    double red[kTileWaves][kObs][4];      // C -> D: per wave and observation: min ll, max ll, sum e^x', sum e^-x'
  };
  double tab[2 * kTabN];
  union {
    T list[kObs][CAP];                    // C, D: the candidates' stored log-likelihoods
    float keys[kObs][kTileSample + 1];    // A, B: the sample, raw = -ll rounded up to f32 (+1: the rows start in different banks)
  };
  double scal[kObs][2];                   // B -> C, D: provisional shift, threshold (both raw)
  unsigned cnt[kObs];                     // candidates seen (not capped)
  unsigned again;                         // D: observations of the group whose list came out too short or too long (bit per observation)
  unsigned next_group;                    // C: the group this workgroup sweeps next (streamed pass: taken from the launch's counter)
  unsigned pad_[2];
  union {
    Sel sel[kTileSelWaves];               // B (histograms of the threshold search), D: scratch of one selection per selecting wave
    Sweep sweep;                          // C, and read at the very start of D
  };
};
static_assert(sizeof(typename TileSmem<double, 640>::Sweep) <= sizeof(ColSmem<CapsSmall>) * kTileSelWaves, "the sweep's scratch lies inside the selection's");
// (the streamed fit kernel's four-wave workgroup: 9 744 bytes static + 30 720 of coefficient scratch at three 64-value blocks,
// tests/test_kernel_resources.py holds it to these 40 960)
static_assert(sizeof(TileSmem<double, kTileCapStream>) + 40960 <= 160 * 1024, "streamed pass: the fit kernel's workgroup must fit beside this one");

template <typename T>
struct CandInTile {  // x = raw - max raw with the reference's single rounding (psis.py:134); the list holds ll = -raw
  const T* list;
  double m;
  ColSmem<CapsSmall>& sm;
  __device__ __forceinline__ double at(unsigned c) const { return (-(double)list[c]) - m; }
  __device__ __forceinline__ unsigned* dump_bin(int lane) const { return &sm.dump_bin[lane]; }
  __device__ __forceinline__ double* dump_slot(int lane) const { return &sm.dump_slot[lane]; }
};

// SYNC: the streamed pass -- the fit kernel runs beside this one and takes each group (= one chunk of kQueueChunk observations)
// as soon as its flag is up: agent-scope whole-line hand-over stores, F.done[group] set once every wave's stores have drained
// (the protocol of the streamed wave kernel, pla_wave.h / pla_fit.h).
template <typename T, bool SYNC>
__global__ __launch_bounds__(kTileThreads, 1) void tile_loo_kernel(TileParams P, FastParams F, int tail_count) {
  static_assert(sizeof(T) == 8, "lane mapping and list size are laid out for 8-byte draws");
  using SMT = TileSmem<T, SYNC ? kTileCapStream : kTileCap>;
  constexpr int kCap = SMT::kCap;
  static_assert(!SYNC || SMT::kObs == kQueueChunk, "a group is a chunk of the streamed fit");
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];
  SMT& sm = *reinterpret_cast<SMT*>(tile_lds);
  constexpr int kObs = SMT::kObs;                   // 16
  constexpr int kSub = kWave / kObs;                // draws per wave load: 4
  constexpr int kStep = kSub * kTileWaves;          // draws per step of the workgroup: 32
  constexpr int kPer = kTileSample / kStep;         // sampled draws per lane: 16
  constexpr int kSelPer = kObs / kTileSelWaves;     // observations a selecting wave takes: 2
  int tid = (int)threadIdx.x;
  const unsigned nblocks = gridDim.x;  // (read once: one scalar instead of a pointer into the hidden arguments kept for the loop)
  for (int j = tid; j < kTabN; j += kTileThreads) exp_table_entry(sm.tab, j);
  if constexpr (SYNC) {
    if (F.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (F.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (F.prio == 3) __builtin_amdgcn_s_setprio(3);
  }
  __syncthreads();
  // (thread, wave and lane numbers are made opaque at the top of every trip of the group loop: the masks and LDS addresses
  // derived from them are then computed where they are used instead of being kept in scalar registers from the top of the
  // kernel -- the rest of the 65 scalars the kernel used to spill into vector lanes)
  int w = __builtin_amdgcn_readfirstlane(tid / kWave);
  int lane = wave_lane();
  int o = lane & (kObs - 1), dsub = lane / kObs;
  const int S = P.n_draws, M = tail_count;
  const int64_t db = P.ld * (int64_t)sizeof(T);      // bytes between consecutive draws
  int64_t step_bytes = db * kStep;
  int nit = S / kStep;                               // whole steps (both laundered per trip of the group loop, see below)
  const int64_t ngroups = (P.n_obs + kObs - 1) / kObs;
  const double INF = pinf();
  const char* tabc = reinterpret_cast<const char*>(sm.tab);

  // ---- D. selection of group gp on its lists in LDS: wave w takes the observations kSelPer w + q -----------------------------
  // `only`: the observations to select for (bit per observation).  `first_try`: an observation whose list came out too short or
  // too long is not handed to the general kernel (which walks this layout at ~1 us per observation, 64 bytes of traffic per
  // draw) but comes round again: its threshold is corrected from the exact count the sweep has just made -- an exponential
  // tail through (threshold, count) and (row maximum, 1), as the long-row kernel's second attempts do (pla_chunked.h) -- and the
  // group is swept once more for it alone.  Returns the observations that want that (0 when first_try is false).
  // (what the selection, the hand-over and the queue need of the launch's arguments is read from the argument block where it is
  // used -- scalar loads from constant memory behind an opaque pointer -- instead of living in scalar registers through the
  // sweep, which has none to spare: the kernel used to keep 65 scalars in vector lanes (v_writelane / v_readlane), a form that
  // came back wrong in round 4's fit kernel; tests/test_kernel_resources.py now holds it to none)
  struct TileArgs {
    TileParams P;
    FastParams F;
    int tail_count;
  };
  typedef const __attribute__((address_space(4))) TileArgs* ArgPtr;
  const auto select_group = [&](const int64_t gp, const unsigned only, const bool first_try) -> unsigned {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only parses this body; it has no constant address space to copy from)
    ArgPtr qp = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(qp));
    const FastParams F = qp->F;  // (shadows the kernel's: loaded here)
#endif
    const int64_t obs0 = gp * kObs;
    if (tid == 0) sm.again = 0u;
    // the sweep's per-wave partial results first, for all of this wave's observations: they lie in the selection's scratch,
    // which the barrier below releases (every lane reads the same entries: broadcasts, and one order of summation)
    double pm[kSelPer], pmn[kSelPer], ps1[kSelPer], ps2[kSelPer];
    if (w < kTileSelWaves) {
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) {
        const int oo = w * kSelPer + q;
        double lmin = INF, lmax = -INF, s1p = 0.0, s2p = 0.0;
#pragma unroll
        for (int k = 0; k < kTileWaves; ++k) {
          const double* rd = sm.sweep.red[k][oo];
          lmin = fmin(lmin, rd[0]);
          lmax = fmax(lmax, rd[1]);
          s1p += rd[2];
          s2p += rd[3];
        }
        pm[q] = -lmin; pmn[q] = -lmax; ps1[q] = s1p; ps2[q] = s2p;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSelPer; ++q) {
      if (w >= kTileSelWaves) break;
      const int oo = w * kSelPer + q;
      const int64_t r = obs0 + oo;
      if (r >= P.n_obs) break;  // (wave-uniform)
      if (!((only >> oo) & 1u)) continue;
      if constexpr ((PLA_TILE_ABLATE & 1) != 0) {
        if (lane == 0) F.ws_s[r * F.ws_sstride + 5] = pm[q] + ps1[q] + (double)sm.cnt[oo];
        continue;
      }
      const double m = uniform_d(pm[q]), mn = uniform_d(pmn[q]);
      const double s1p = uniform_d(ps1[q]), s2p = uniform_d(ps2[q]);
      const double mp = uniform_d(sm.scal[oo][0]), t_raw = uniform_d(sm.scal[oo][1]);
      const int ncand = __builtin_amdgcn_readfirstlane((int)sm.cnt[oo]);
      const double R = m - mn, delta = m - mp;  // (delta: the row's maximum against the sample's, rounded up to f32)
      const bool bad_row = !(R < kWaveMaxRange) || !(fabs(s1p) < INF) || !(fabs(s2p) < INF) || !(fabs(delta) < kWaveMaxRange);
      const bool bad_count = ncand < M + 1 || ncand > kCap;
      if (first_try && bad_count && !bad_row && ncand >= 2 && m > t_raw) {
        // count(x) ~ e^(-lambda (max - x)) through (t_raw, ncand) and (m, 1): the threshold with the middle of the list's
        // range above it
        const float ratio = __logf(0.5f * (float)(M + 1 + kCap)) / __logf((float)ncand);
        if (lane == 0) {
          sm.scal[oo][1] = m - (m - t_raw) * (double)ratio;
          atomicOr(&sm.again, 1u << oo);
        }
        continue;
      }
      bool slow = bad_row || bad_count;
      typename SMT::Sel& ss = sm.sel[w];
      if (!slow) {
        wave_sync();  // the previous observation is done with the scratch
        {
          const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int j = 0; j < kWaveBins / (4 * kWave); ++j) *reinterpret_cast<uint4*>(&ss.hist[4 * (lane + kWave * j)]) = z4;
        }
        double magic = kMagic, c256 = kC256;
        // histogram origin: every candidate has raw >= t_raw, so x >= t_raw - m up to the two roundings: two keys of margin
        const int k1 = __double2loint(fma(t_raw - m, c256, magic)) - 2;
        const int span = -k1;
        const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
        // the sums about the true shift: e^x = e^x' e^-(m - m'), e^-x = e^-x' e^(m - m')
        const double s1 = lane == 0 ? s1p * exp_tab(-delta, sm.tab) : 0.0;
        const double s2 = lane == 0 ? s2p * exp_tab(delta, sm.tab) : 0.0;
        const CandInTile<T> src{sm.list[oo], m, ss};
        wave_sync();
        wave_select_split<typename SMT::Sel, SMT, (CapsSmall::kMaxTail + 63) / 64, CandInTile<T>, SYNC>(
            F, ss, sm, r, lane, M, m, mn, s1, s2, (unsigned)ncand, k1, sh, magic, c256, slow, src);
      }
      // (a scalar branch: the wait for the counter's old value must not lie on the path of the observations that stay -- it
      // would also wait for the next group's loads, which are in flight through this selection)
      if (__builtin_amdgcn_readfirstlane((int)slow) != 0 && PLA_TILE_ABLATE == 0) {
        // (tail length -1: on the list for the general kernel, nothing for the fit kernel)
        ws_store_scalars<SYNC>(F, r, lane, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0);
        if (lane == 0) {
          const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
          F.slow_list[idx] = (unsigned)r + F.slow_base;
        }
      }
    }
    // the lists and the scratch are free for the next group from here on; streamed pass: every hand-over store of the group
    // has drained (each wave waits for its own, then the barrier), and the fit kernel may have the group
    if constexpr (SYNC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned again = (unsigned)__builtin_amdgcn_readfirstlane((int)sm.again);
    if constexpr (SYNC) {  // (a group that is swept again is the fit kernel's when that is over)
      if (tid == 0 && again == 0u) __hip_atomic_store(&F.done[gp], (unsigned)kQueueChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (all rows of the chunk: pla_fast.h, kQueueUnit)
    }
    return again;
  };
  // Every wave sweeps the row's steps cyclically from a start of its own, w / 8 of the way through the row.  The first 16
  // steps a wave loads -- the head of its ring -- are then its part of the SAMPLE the thresholds come from: 16 clusters of four
  // consecutive draws, 32 draws apart, from eight windows spread evenly over the row (at S = 4000 the eight windows tile the
  // row: a cluster every 32 draws; every chain of a chain-major stack contributes).  Nothing is loaded for the sample alone
  // (a sample of its own, read again by the sweep, was 11 % of the kernel's loads and of its HBM traffic).
  static_assert(kPer <= PLA_TILE_RING, "the sample is the head of the ring");
  int start_w = 0, ahead0 = 0;  // first step of this wave's sweep / the step of visit R (set at the top of every trip)
  // One trip of the loop: the sample loads of group g are ISSUED, the selection of the group before it runs on the lists in LDS
  // while they fly, then the sample is turned into thresholds and the group is swept.  (One place of issue for the sample; the
  // loop is entered with no group behind and left with none ahead.)
  int64_t g = blockIdx.x, prev = -1, resume = -1;
  unsigned prev_only = 0xffffu;  // the observations of `prev` to select for
  bool prev_first = true;        // ... for the first time
  unsigned redo = 0u;            // non-zero: group g is swept a second time, for these observations
  for (;;) {
    asm volatile("" : "+v"(tid), "+v"(lane));
    asm volatile("" : "+s"(w), "+s"(nit), "+s"(step_bytes));
    o = lane & (kObs - 1);
    dsub = lane / kObs;
    start_w = (int)(((unsigned)w * (unsigned)nit) / (unsigned)kTileWaves);
    ahead0 = PLA_TILE_RING + start_w;
    ahead0 = ahead0 >= nit ? ahead0 - nit : ahead0;  // (nit >= the ring: tile_supported asks for 512 draws)
    const bool have = g < ngroups;
    const int64_t obs0 = g * kObs;
    // (lanes past the last observation of the launch re-read the last one; nothing of theirs is selected)
    const int oc = (obs0 + o < P.n_obs) ? o : (int)(P.n_obs - 1 - obs0);
    // this lane inside a step: its observation and its draw of the wave's kSub (32-bit: ld < 2^32 / (8 kSub), checked by the launcher)
    const int voff = (int)((unsigned)(oc * (int)sizeof(T)) + (unsigned)dsub * (unsigned)db);
    // the wave's first piece of step 0 (scalar)
    const char* gbase = reinterpret_cast<const char*>(P.in) + obs0 * (int64_t)sizeof(T) + (int64_t)(kSub * w) * db;
    const auto load_step = [&](int step) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(gbase + (int64_t)step * step_bytes), 0, (int)0xfffffff0u, 0x00020000);
      return col_load<T>(rs, voff, 0);
    };
    // ---- A. the first R steps of this wave's sweep, issued one group ahead: in flight through the selection of the group
    // before, so that the threshold search finds its sample and the sweep a full pipeline.  (Visits past the row's last step
    // come round to its first steps again: valid memory, never looked at.)
    constexpr int R = PLA_TILE_RING;
    T ring[R];
    const char* const rowend = gbase + (int64_t)nit * step_bytes;
    int rs_records = (int)0xfffffff0u, rs_flags = 0x00020000;
    asm volatile("" : "+s"(rs_records), "+s"(rs_flags));
    if (have) {
      const char* at = gbase + (int64_t)start_w * step_bytes;
#pragma unroll
      for (int u = 0; u < R; ++u) {
        ring[u] = col_load<T>(__builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(at), 0, rs_records, rs_flags), voff, 0);
        at += step_bytes;
        at = at == rowend ? gbase : at;
      }
    }
    if (prev >= 0) {  // (ends on a barrier: the lists of the group before are free for this group's sample)
      const unsigned again = select_group(prev, prev_only, prev_first);
      if (again != 0u) {  // once more through `prev` (the loads just issued are dropped: a few groups in a hundred)
        resume = g;
        g = prev;
        prev = -1;
        redo = again;
        continue;
      }
    }
    if (!have) break;
    if (redo != 0u) {
      // second sweep of a group: the corrected thresholds are in place, everybody else's is out of reach
      if (tid < kObs) {
        if (!((redo >> tid) & 1u)) sm.scal[tid][1] = INF;
        sm.cnt[tid] = 0u;
      }
    } else if constexpr (!(PLA_TILE_ABLATE & 2)) {
#pragma unroll
      for (int j = 0; j < kPer; ++j)  // rounded up: the threshold may only err towards FEWER candidates by what one float ulp is worth
        sm.keys[o][(w * kSub + dsub) * kPer + j] = __double2float_ru(-(double)ring[j]);
      __syncthreads();
      // ---- B. threshold and provisional shift: a wave per observation, kSelPer observations side by side ------------------------
      // The sample's ks-th largest value to 1/4096 of the sample's range, by counting: a 64-bin histogram of the eight keys a
      // lane holds (one bin per lane, LDS atomics), a suffix sum over the lanes, the bin that holds the ks-th largest -- and the
      // same again inside that bin.  (Bisection on ballot counts, 12 halvings x 8 compares per observation, was a fifth of
      // the instructions of the whole sweep.)  The threshold is the lower edge of the final sub-bin: ks .. ks + (members of
      // that sub-bin) - 1 of the sample lie at or above it.
      if (w < kTileSelWaves) {
      constexpr int KX = kTileSample / kWave;
      float kx[kSelPer][KX], hi[kSelPer];
      double hmax[kSelPer];
      unsigned* const hh = sm.sel[w].hist;  // (the selection's scratch is idle here) [level][observation][64] + 64 dump words
      {
        const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
        static_assert(kWaveBins >= 4 * kWave * kSelPer / 2 + kWave, "histograms of the threshold search fit the selection's");
        *reinterpret_cast<uint4*>(&hh[4 * lane]) = z4;  // 256 words: two levels x two observations
      }
      float lo1[kSelPer], w1[kSelPer];
      int bin[kSelPer][KX];
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) {
        float lmin = __builtin_inff(), lmax = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < KX; ++k) {
          kx[q][k] = sm.keys[w * kSelPer + q][lane + kWave * k];
          lmin = fminf(lmin, kx[q][k]);
          lmax = fmaxf(lmax, kx[q][k]);
        }
        double nlmin;
        wave_all2<R_MAX>((double)lmax, -(double)lmin, hmax[q], nlmin);
        lo1[q] = (float)(-nlmin);
        w1[q] = ((float)hmax[q] - lo1[q]) * (1.0f / 64.0f);
      }
      wave_sync();
      const auto level = [&](const int lv, const float (&lo)[kSelPer], const float (&wd)[kSelPer], const int (&want)[kSelPer],
                             const int (&only)[kSelPer], int (&bstar)[kSelPer], int (&need)[kSelPer]) {
#pragma unroll
        for (int q = 0; q < kSelPer; ++q) {
          const float sc = 1.0f / wd[q];
          unsigned* const h = hh + (lv * kSelPer + q) * kWave;
#pragma unroll
          for (int k = 0; k < KX; ++k) {
            int b = (int)((kx[q][k] - lo[q]) * sc);
            b = b < 0 ? 0 : (b > kWave - 1 ? kWave - 1 : b);
            const bool in = lv == 0 || bin[q][k] == only[q];
            if (lv == 0) bin[q][k] = b;
            atomicAdd(in ? &h[b] : &hh[2 * kSelPer * kWave + lane], 1u);
          }
        }
        wave_sync();
#pragma unroll
        for (int q = 0; q < kSelPer; ++q) {
          const unsigned c = hh[(lv * kSelPer + q) * kWave + lane];
          unsigned pre = c;  // inclusive prefix over the lanes
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xF, 0xF, true);     // row_shr:1
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xF, 0xF, true);     // row_shr:2
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x113, 0xF, 0xF, true);     // row_shr:3
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x114, 0xF, 0xE, false);  // row_shr:4, banks 1-3
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x118, 0xF, 0xC, false);  // row_shr:8, banks 2-3
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x142, 0xA, 0xF, false);  // row_bcast:15
          pre += (unsigned)__builtin_amdgcn_update_dpp(0, (int)pre, 0x143, 0xC, 0xF, false);  // row_bcast:31
          const unsigned all = (unsigned)__builtin_amdgcn_readlane((int)pre, kWave - 1);
          const int above = (int)(all - pre);  // members of the higher bins
          // the bin that holds the want-th largest; none does when fewer than `want` are counted at all (then: the lowest bin)
          const unsigned long long who = __ballot(above < want[q] && want[q] <= above + (int)c);
          const int src = who ? __ffsll((long long)who) - 1 : 0;
          bstar[q] = src;
          need[q] = want[q] - __builtin_amdgcn_readlane(above, src);
        }
      };
      int want1[kSelPer], none[kSelPer], b1[kSelPer], need1[kSelPer], b2[kSelPer], need2[kSelPer];
      float lo2[kSelPer], w2[kSelPer];
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) {
        want1[q] = P.ks;
        none[q] = 0;
      }
      level(0, lo1, w1, want1, none, b1, need1);
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) {
        lo2[q] = fmaf((float)b1[q], w1[q], lo1[q]);
        w2[q] = w1[q] * (1.0f / 64.0f);
      }
      level(1, lo2, w2, need1, b1, b2, need2);
#pragma unroll
      for (int q = 0; q < kSelPer; ++q) hi[q] = fmaf((float)b2[q], w2[q], lo2[q]);
      wave_sync();
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kSelPer; ++q) {
          sm.scal[w * kSelPer + q][0] = hmax[q];
          sm.scal[w * kSelPer + q][1] = (double)hi[q];
          sm.cnt[w * kSelPer + q] = 0u;
        }
      }
      }
    } else if (tid < kObs) {
      sm.scal[tid][0] = 0.0;
      sm.scal[tid][1] = 1e300;
      sm.cnt[tid] = 0u;
    }
    __syncthreads();

    // The groups after a workgroup's first come from a counter of the launch (F.queue, zero at launch), asked for
    // at the start of the sweep they follow -- a whole sweep ahead of their use.  With the groups dealt out by blockIdx alone the
    // streamed launch would depend on every workgroup being resident: the fit kernel's workgroups share the CUs, and one of this kernel's
    // that found no room beside two of them would leave its groups unswept while they wait for those very groups.  This way
    // a displaced workgroup's groups go to its neighbours.
    // (Back to back, the same counter evens out what the second sweeps and the last round of groups leave uneven: 9.1-9.4 ->
    // 8.8-8.9 ms on C3 when it came in.)
    if (tid == 0 && redo == 0u) {
      ArgPtr qp = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(qp));
      sm.next_group = nblocks + atomicAdd(qp->F.queue, 1u);
    }
    // ---- C. the sweep ---------------------------------------------------------------------------------------------------------
    const double nmp = -sm.scal[o][0], nt_raw = -sm.scal[o][1];
    double nmn = INF, nmx = -INF, s1 = 0.0, s2 = 0.0;  // min / max of ll = -(max / min of raw)
    T* const mylist = sm.list[o];
    unsigned* const mycnt = &sm.cnt[o];
    unsigned* const mydumpc = &sm.sweep.dumpc[tid];
    *mydumpc = 0x80000000u;  // (this thread's own word, in scratch the threshold search and the selection have used since)
    T* const mydumpv = &sm.sweep.dumpv[tid];
    // the constants of the sweep live in registers for its whole length (MachineLICM is off for this library: the compiler
    // would otherwise re-materialise each of them with s_mov per use -- ten scalar instructions per draw)
    int c4096 = 4096, cm4096 = -4096, four = 4;
    double c256 = kC256, magic = kMagic, nmagic = -kMagic, nl256 = -kLn2_256, c6 = 1.66666666666666666667e-01;
    asm volatile("" : "+s"(c4096), "+s"(cm4096), "+s"(c256), "+s"(nl256), "+s"(c6));
    asm volatile("" : "+v"(four), "+v"(magic), "+v"(nmagic));
    // Software pipeline, kPF draws deep and carried from batch to batch: stage A of a draw (shift, range reduction, table
    // read, the request for a list slot) is issued kPF draws before its stage B (polynomial, accumulation, the store to the
    // slot), so that the LDS round trips of the table read and of the counter are covered by the arithmetic in between.
    constexpr int kPF = PLA_TILE_PF;
    double px[kPF], pt[kPF], pll[kPF];
    int4 ptt[kPF];
    unsigned ppos[kPF];
    const auto stage_a = [&](const double ll, const int sl) {
      nmn = vmin_nc(nmn, ll);
      if constexpr ((PLA_TILE_ABLATE & 4) != 0) return;                          // (a NaN draw is ignored here and poisons the sums: general kernel)
      nmx = vmax_nc<false>(nmx, ll);
      const double x = nmp - ll;                       // raw - m': psis.py:134 about the provisional shift
      const double t = fma(x, c256, magic);
      const int k = __double2loint(t);
      ptt[sl] = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));
      // candidate: a slot of the observation's list from its counter (32 lanes in 8 waves append to one list); everybody else
      // counts on a private counter that starts past the end of any list, and stores to a private slot: no control flow
      // (as `if (candidate)` every draw becomes a region of its own and the pipeline's registers spill: 256 + 148)
      const bool cand = ll <= nt_raw;                  // raw >= threshold
      if constexpr ((PLA_TILE_ABLATE & 8) != 0) {
        nmx = vmax_nc<false>(nmx, cand ? 1.0 : 2.0);
      } else if constexpr ((PLA_TILE_ABLATE & 64) != 0) {
        ppos[sl] = cand ? 0xffffffffu : 0xfffffff0u;
      } else {
        ppos[sl] = __hip_atomic_fetch_add(cand ? mycnt : mydumpc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      px[sl] = x;
      pt[sl] = t;
      // (an explicit copy: the ring slot this draw came from is dead from here on, so that the load issued right behind this
      // stage lands in the slot's own registers -- left to itself the compiler keeps the draw where it is, loads into spare
      // registers and moves them home at the end of the round behind an s_waitcnt vmcnt(1) that drains the whole ring)
      asm("v_mov_b64 %0, %1" : "=v"(pll[sl]) : "v"(ll));
    };
    const auto stage_b = [&](const int sl) {
      if constexpr ((PLA_TILE_ABLATE & 4) != 0) return;
      if constexpr ((PLA_TILE_ABLATE & 16) != 0) {
        s1 += px[sl];
        if constexpr (!(PLA_TILE_ABLATE & 8)) *(ppos[sl] < (unsigned)kCap ? &mylist[ppos[sl]] : mydumpv) = (T)pll[sl];
        return;
      }
      const double x = px[sl], t = pt[sl];
      const int k = __double2loint(t);
      const double rr = fma(t + nmagic, nl256, x);
      const double r2 = rr * rr;
      const double E = fma(r2, 0.5, 1.0);              // cosh rr to 1.5e-13 (as in the wave kernel's sweep)
      const double O = fma(c6, r2, 1.0);
      s1 = fma(__hiloint2double(mad_i24(k, c4096, ptt[sl].y), ptt[sl].x), fma(rr, O, E), s1);
      s2 = fma(__hiloint2double(mad_i24(k, cm4096, ptt[sl].w), ptt[sl].z), fma(-rr, O, E), s2);
      if constexpr (!(PLA_TILE_ABLATE & 8)) {
        if constexpr ((PLA_TILE_ABLATE & 32) != 0) {
          nmx = vmax_nc<false>(nmx, ppos[sl] < (unsigned)kCap ? 1.0 : 2.0);
        } else {
          *(ppos[sl] < (unsigned)kCap ? &mylist[ppos[sl]] : mydumpv) = (T)pll[sl];
        }
      }
    };
    const auto one = [&](double ll) {  // (outside the batches: the few draws behind them)
      stage_a(ll, 0);
      stage_b(0);
    };
    {
      // The stream: a ring of R steps per lane.  The load of step i + R is issued into the slot of step i as soon as that
      // draw has entered the pipeline, so R - kPF loads per lane are in flight at every moment (two half-rings that are
      // fetched and consumed in turn keep only one of them in flight while the other is computed on: at two waves per SIMD
      // that left the sweep waiting on HBM latency, 48 KB per CU outstanding where ~100 KB are needed).
      static_assert(R % kPF == 0, "pipeline slots are assigned at compile time");
      const int nmain = (nit / R) * R;
      const char* ahead = gbase + (int64_t)ahead0 * step_bytes;   // where visit base + R + u lies
      // (the pipeline starts on kPF draws that count nothing: table entry 0 x 2^0 = 0.0 times a polynomial of 1, stored to the dump slot)
#pragma unroll
      for (int u = 0; u < kPF; ++u) {
        px[u] = 0.0;
        pt[u] = kMagic;
        pll[u] = 0.0;
        ptt[u] = make_int4(0, 0, 0, 0);
        ppos[u] = 0xffffffffu;
      }
#pragma unroll 1
      for (int base = 0; base < nmain; base += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          stage_b(u % kPF);
          stage_a((double)ring[u], u % kPF);
          // the visit R ahead, through a running pointer that comes round at the row's end
          asm volatile("" : "+s"(ahead));
          ring[u] = col_load<T>(__builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ahead), 0, rs_records, rs_flags), voff, 0);
          ahead += step_bytes;
          ahead = ahead == rowend ? gbase : ahead;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < kPF; ++u) stage_b(u);
      // the steps behind the last whole round (already in the ring), and the draws behind the last whole step
      int left_steps = nit - nmain;
      // (opaque per group: the fifteen tests below are then made here, scalar compare and branch, instead of being kept as
      // fifteen 64-bit masks from the top of the kernel -- thirty of the scalars the kernel used to spill into vector lanes)
      asm volatile("" : "+s"(left_steps));
      if (left_steps > 0) {
#pragma unroll
        for (int u = 0; u < R - 1; ++u)
          if (u < left_steps) one((double)ring[u]);
      }
      if (nit * kStep + kSub * w + dsub < S) one((double)load_step(nit));
    }
    // ---- the 32 lanes of an observation: the four of a wave here, the eight waves in LDS ---------------------------------------
#pragma unroll
    for (int off = kObs; off < kWave; off *= 2) {
      nmn = fmin(nmn, __shfl_xor(nmn, off));
      nmx = fmax(nmx, __shfl_xor(nmx, off));
      s1 += __shfl_xor(s1, off);
      s2 += __shfl_xor(s2, off);
    }
    if (dsub == 0) {
      double* rd = sm.sweep.red[w][o];
      rd[0] = nmn; rd[1] = nmx; rd[2] = s1; rd[3] = s2;
    }
    __syncthreads();
    prev = g;
    prev_only = redo != 0u ? redo : 0xffffu;
    prev_first = redo == 0u;
    g = redo != 0u ? resume : (int64_t)sm.next_group;  // (written before the barrier above; a second sweep keeps the group it had in hand)
    redo = 0u;
  }
}

}  // namespace pla
