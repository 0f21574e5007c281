// Chunked wave-per-observation LOO kernel: rows longer than one register chunk (S > 4096) and/or tail
// counts up to 512 (small reff, S = 20 000).  Same pipeline as pla_wave.h, still ONE read of the row:
//
//   * the row passes through the 64 register slots per lane in chunks of 4096 draws; the loads of the
//     next chunk (or of the next row's first chunk) stream into the registers the sweep has consumed;
//   * nothing in the sweep needs the row maximum any more: the exponentials are taken relative to a
//     PROVISIONAL shift m' (the maximum of the first chunk) and the two sums are rescaled by
//     e^(-+(m - m')) once the true maximum m is known (|x'| <= R < 690 keeps everything finite);
//   * the candidate threshold comes from the first chunk alone (the draws are exchangeable, so its
//     group maxima estimate the same quantile of the whole row); candidates are kept as the raw
//     input values and turned into x = raw - m exactly as the reference computes it (one rounding)
//     after the last chunk;
//   * per chunk: pad fix-up, max / min, sweep; only the last chunk can have pads, and by then the
//     running minimum is final, so the pad value (smallest x of the row) and its exactly known
//     contribution are handled as in the one-chunk kernel.
//
// Everything after the sweep is wave_back() of pla_wave.h.  LDS capacities: CapsBig (2 waves per
// workgroup, 4 per CU).  Shapes: unit draw stride, 16-byte aligned rows, every chunk >= 256 draws.
#pragma once

#include "pla_wave.h"

namespace pla {

constexpr int kChunkDraws = kWave * kWaveSlots;  // 4096

// A long row is read ONCE, so its threshold can only know the first chunk -- exact for that chunk, blind to what later chains
// of a chain-major stack do differently.  A row that ends with too few or too many draws above it is NOT handed to the general
// kernel (twelve times the cost) but comes round a second time: the kernel's row loop schedules it again behind the row that
// is already streaming in (one place where loads are issued, nothing waits), with the row's true maximum as the shift and a
// threshold corrected by what the first attempt counted (ChunkRetry).
struct ChunkRetry {
  double t_raw;    // threshold on the raw scale
  double m_raw;    // the row's maximum, known from the first attempt
  int64_t row;     // -1: none
  double t_prev;   // the attempt before: its threshold (raw scale) ...
  double ln_prev;  // ... and the log of the number of draws that were above it
  double lambda0;  // tail rate seen by the first attempt (first chunk)
  int attempt;     // 0: a fresh row; kChunkAttempts - 1 is the last one
};
constexpr int kChunkAttempts = 4;

template <typename T, int VEC, typename SM, typename TB, bool SPLIT = false, bool LW = false>
__device__ PLA_ROW_INLINE void wave_loo_row_chunked(const RowsParams& P, const FastParams& F, SM& sm, const TB& tb,
                                                    const int64_t r, T (&v)[kWaveSlots], const T* row, const T* rp_next,
                                                    const ChunkRetry& now, ChunkRetry& want) {
  // LW with SPLIT (weights of long rows, round 4): only the SIGN convention of weights mode -- raw = the input, not its negative
  // -- and everything else as in the LOO split pass: no draw indices, no output pass here (the fit kernel and lw_output_kernel
  // follow, pla_lwout.h).  LWF: the fused weights mode of rounds 2-3.
  constexpr bool LWF = LW && !SPLIT;
  const bool second = now.attempt > 0;
  const double t_second = now.t_raw, m_second = now.m_raw;
  constexpr int EPT = kWaveSlots;
  constexpr int NQ = EPT / VEC;
  constexpr int kCand = SM::Caps::kCand;
  // (opaque to the optimiser: otherwise every lane-derived mask of the later phases is hoisted out of the ROW loop into
  // scalar registers, which that loop does not have -- they were being spilled to vector lanes)
  int lane = wave_lane();
  asm volatile("" : "+v"(lane));
  const int S = __builtin_amdgcn_readfirstlane(P.n_draws);
  const int M = __builtin_amdgcn_readfirstlane(P.tail_count);
  const int gsz = __builtin_amdgcn_readfirstlane(F.gsz);
  const int kq = __builtin_amdgcn_readfirstlane(F.kq);
  const int sbits = __builtin_amdgcn_readfirstlane(F.sample_bits);
  constexpr int dbgs = 0;
  const int mestM = __builtin_amdgcn_readfirstlane(F.mest_M);
  const double logS = uniform_d(F.log_S);
  const double INF = pinf();
  const int nch = (S + kChunkDraws - 1) / kChunkDraws;

  double magic = kMagic, c256 = kC256;
  asm volatile("" : "+v"(magic));
  asm volatile("" : "+s"(c256));
  const auto key_of = [&](double xx) { return __double2loint(fma(xx, c256, magic)); };

  // state carried across the chunks of the row
  double m_run = -INF, mn_run = INF;  // running max / min of raw = -ll (wave-uniform)
  double mp = 0.0;                    // provisional shift: max raw of chunk 0
  double t1p = 0.0;                   // candidate threshold relative to mp
  double s1 = 0.0, s2 = 0.0;          // per-lane sums of e^x', e^-x'  (x' = raw - mp)
  bool slow = false;
  const unsigned cand0 = lds_addr(sm.cand);
  const unsigned dump8 = cand0 + (unsigned)(kCand + kWave + lane) * 8u;
  const unsigned lim8 = cand0 + 8u * kCand;
  unsigned next8 = cand0, base8 = cand0;
  const char* tabc = reinterpret_cast<const char*>(tb.tab);
  unsigned c_first = 0;   // draws of the first chunk above the threshold

  wave_sync();  // previous row is done with the LDS scratch
  {
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < kWaveBins / (4 * kWave); ++i) *reinterpret_cast<uint4*>(&sm.hist[4 * (lane + kWave * i)]) = z4;
  }

#pragma unroll 1
  for (int ch = 0; ch < nch; ++ch) {
    const int Sc = (S - ch * kChunkDraws < kChunkDraws) ? S - ch * kChunkDraws : kChunkDraws;  // draws in this chunk
    const int nvec = Sc / VEC, qfull = nvec / kWave, qrem = nvec - qfull * kWave;
    const bool last = (ch == nch - 1);
    // ---- pad fix-up (copies of the lane's first vector), max / min of the chunk --------------------
    pad_tail<T, VEC, NQ - 1, false>(v, qfull, qrem, (T)0);
    double mx, mn, gs;
    row_stats<T, VEC, LW>(v, gsz, sbits, mx, mn, gs);  // (gs: the threshold sample, used for chunk 0 only)
    double mc, nmnc;
    wave_all2<R_MAX>(mx, -mn, mc, nmnc);  // min = -max(-.)
    const double mnc = -nmnc;
    m_run = fmax(m_run, mc);
    mn_run = fmin(mn_run, mnc);
    if (ch == 0) {
      // (a row that is its own last chunk has pads: make them the row minimum before anything is counted)
      if (last) pad_tail<T, VEC, NQ - 1, true>(v, qfull, qrem, LW ? (T)mn_run : (T)(-mn_run));
      // speculative threshold from the first chunk's group maxima (see pla_wave.h); shift = its maximum
      mp = mc;
      double hi = mc;
      if (!second) {
        double lo = wave_all<R_MIN>(gs);
#pragma unroll 1
        for (int it = 0; it < 9; ++it) {
          const double mid = 0.5 * (lo + hi);
          const int below = __popcll(__ballot(gs < mid));
          if (below >= kq) hi = mid; else lo = mid;
        }
        if (mc - mnc < kWaveMaxRange) {
          // (the exact counts look at this first chunk only: the rest of the row has not been read yet)
          if (!wave_threshold_check<T, VEC, LW>(v, F, mnc, mc, hi)) slow = true;
        }
      } else {
        // second attempt: the row's maximum is known (the shift is final from the start) and so is the threshold
        mp = fmax(mc, m_second);
        hi = t_second;
      }
      t1p = hi - mp;
      if (!(t1p < 0.0)) slow = true;
    }
    if (ch == 1) c_first = (next8 - cand0) >> 3;
    // a non-finite maximum (inf / NaN draws) or a range that may overflow the sums: general kernel.
    // The chunk is still swept (clamped shift) so that the streaming of the following chunk goes on.
    if (!(m_run - mn_run < kWaveMaxRange)) slow = true;
    const double shift = slow ? m_run : mp;  // any finite-or-not value will do once the row is lost
    // pads (last chunk only): the smallest raw value of the row, now final
    const double xpad = mn_run - mp;
    if (last) {
      if (key_of(xpad) >= key_of(t1p)) slow = true;  // pads would be counted as candidates
      pad_tail<T, VEC, NQ - 1, true>(v, qfull, qrem, LW ? (T)mn_run : (T)(-mn_run));
    }
    // ---- sweep of the chunk (pla_wave.h, section 2) -------------------------------------------------
    double nl256 = -kLn2_256, c6 = 1.66666666666666666667e-01;
    int c4096 = 4096, cm4096 = -4096;
    asm volatile("" : "+s"(nl256), "+s"(c6), "+s"(c4096), "+s"(cm4096));
    int four = 4;
    asm volatile("" : "+v"(four));
    // what streams in behind the sweep: the next chunk of this row, or the first chunk of the next row
    // (weights mode: the row's own first chunk again, for the output pass)
    const T* rp_stream = last ? (LWF ? row : rp_next) : row + (int64_t)(ch + 1) * kChunkDraws;
    int bytes_stream = 0;
    if (rp_stream) {
      const int left = last ? S : S - (ch + 1) * kChunkDraws;
      bytes_stream = (left < kChunkDraws ? left : kChunkDraws) * (int)sizeof(T);
    }
    const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(rp_stream ? rp_stream : row), 0, bytes_stream, 0x00020000);
    int id_base = VEC * lane + ch * kChunkDraws;
    asm volatile("" : "+v"(id_base));
    constexpr int kPF = 3;
    double px[kPF], pt[kPF];
    int4 ptt[kPF];
#pragma unroll
    for (int i = 0; i < EPT + kPF; ++i) {
      if (i >= kPF) {  // stage B of draw i - kPF
        const int sl = (i - kPF) % kPF;
        const double x = px[sl], t = pt[sl];
        const int k = __double2loint(t);
        const double rr = fma(t - magic, nl256, x);
        const double r2 = rr * rr;
        const double E = fma(r2, 0.5, 1.0);
        const double O = fma(c6, r2, 1.0);
        s1 = fma(__hiloint2double(mad_i24(k, c4096, ptt[sl].y), ptt[sl].x), fma(rr, O, E), s1);
        s2 = fma(__hiloint2double(mad_i24(k, cm4096, ptt[sl].w), ptt[sl].z), fma(-rr, O, E), s2);
        if ((i & 3) == 3) asm volatile("" : "+v"(s1), "+v"(s2));
      }
      if (i < EPT) {  // stage A of draw i
        const int sl = i % kPF;
        const double x = LW ? (double)v[i] - shift : (-(double)v[i]) - shift;
        const double t = fma(x, c256, magic);
        const int k = __double2loint(t);
        px[sl] = x;
        pt[sl] = t;
        ptt[sl] = *reinterpret_cast<const int4*>(tabc + byte0_shl(k, four));
        const bool cand = x >= t1p;
        const unsigned long long cm = __ballot(cand);
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
        unsigned pos8 = (rank << 3) + base8;
        asm("" : "+v"(pos8));
        lds_store(cand ? pos8 : dump8, (double)v[i]);  // the INPUT value: x = -value - m follows after the last chunk
        if constexpr (LWF)  // draw index of the candidate: 4096 ch + VEC * (lane + 64 q) + e for slot i = q VEC + e
          sm.ids[cand ? ((pos8 - cand0) >> 3) : (unsigned)(kCand + kWave + lane)] =
              (unsigned short)(id_base + (VEC * kWave * (i / VEC) + i % VEC));
        {
          const unsigned pc = (unsigned)__popcll(cm);
          asm("s_lshl3_add_u32 %0, %1, %0" : "+s"(next8) : "s"(pc) : "scc");
        }
        base8 = next8 < lim8 ? next8 : lim8;
        if ((i % VEC) == VEC - 1) issue_row_vector<T, VEC>(v, rs_next, i / VEC);
      }
    }
    if (last) {  // remove the pads' contribution (same code path, so it cancels to rounding)
      const double x = xpad;
      const double t = fma(x, kC256, magic);
      const int k = __double2loint(t);
      const double rr = fma(t - magic, -kLn2_256, x);
      const int4 tt = *reinterpret_cast<const int4*>(tabc + 16 * (k & 255));
      const double r2 = rr * rr;
      const double E = fma(r2, 0.5, 1.0);
      const double O = fma(1.66666666666666666667e-01, r2, 1.0);
      const double npad = (double)((NQ - qfull) * VEC - ((lane < qrem) ? VEC : 0));
      s1 = fma(-npad * __hiloint2double(tt.y + (k << 12), tt.x), fma(rr, O, E), s1);
      s2 = fma(-npad * __hiloint2double(tt.w - (k << 12), tt.z), fma(-rr, O, E), s2);
    }
  }

  // ---- the row maximum is known: candidates -> x, sums -> true shift ---------------------------------
  const double m = m_run, mn = mn_run, R = m - mn;
  const double delta = m - mp;  // >= 0
  const unsigned ncand = (next8 - cand0) >> 3;  // (the true count, also past the list's capacity)
  double khat = INF, loo = 0.0, lppd = 0.0;
  wave_sync();
  if (nch == 1) c_first = ncand;
  if (!slow && ((int)ncand < M + 1 || ncand > (unsigned)kCand)) {
    slow = true;
    if constexpr (!LWF) {
      if (now.attempt + 1 < kChunkAttempts && F.retry_target > 0 && c_first >= 8u && ncand >= 8u && R < kWaveMaxRange) {
        // The next threshold, from EXACT counts of this row (dependence between neighbouring draws does not bias them) and an
        // exponential tail, log count(t) = a - lambda t:
        //   first return   the threshold t (relative to the first chunk's maximum, where one draw of that chunk lies) had c_first
        //                  draws of the first chunk above it: lambda = log(c_first) / -t;
        //   later returns  the secant through the last two attempts (t', n') and (t, n) -- the row's own tail, whatever its
        //                  chains do (a chain that lies above the threshold as a whole saturates the first estimate).
        //   a chain above the threshold as a whole (the count is several times the list, or did not move between two attempts):
        //                  the threshold is anchored at the row's maximum instead, ln(target) / lambda below it -- where an
        //                  exponential tail that ends in that maximum holds `target` draws.
        const double t_now = mp + t1p, ln_now = log_fast((double)ncand), ln_tgt = log_fast((double)F.retry_target);
        const double lambda0 = now.attempt > 0 ? now.lambda0 : log_fast((double)c_first) / -t1p;
        double lambda = lambda0;
        bool moved = true;
        if (now.attempt > 0 && now.t_prev != t_now) {
          const double sec = (now.ln_prev - ln_now) / (t_now - now.t_prev);
          moved = fabs(now.ln_prev - ln_now) > 0.05;
          if (sec > 0.0 && moved) lambda = sec;
        }
        double t_new = t_now + (ln_now - ln_tgt) / lambda;  // up when there were too many
        if (ncand > 3u * (unsigned)kCand || !moved) t_new = fmax(t_new, m - ln_tgt / lambda0);
        t_new = fmin(fmax(t_new, t_now - 3.0 * -t1p), m - 1e-3 * (m - mn));  // (stays below the row's maximum)
        want.t_raw = t_new;
        want.lambda0 = lambda0;
        want.m_raw = m;
        want.row = r;
        want.t_prev = t_now;
        want.ln_prev = ln_now;
        want.attempt = now.attempt + 1;
      }
    }
  }
  if (!slow) {
    for (unsigned c = lane; c < ncand; c += kWave) sm.cand[c] = LW ? sm.cand[c] - m : (-sm.cand[c]) - m;  // psis.py:134, one rounding
    s1 *= exp_tab(-delta, tb.tab);  // e^x = e^x' e^-(m - m');  s2 is rescaled in log space (lppd_shift)
    // histogram origin one key below the threshold: x and the threshold were rounded on different paths
    const int k1 = key_of(t1p - delta) - 1;
    const int span = -k1;
    const int sh = (span >> 9) ? (32 - __builtin_clz((unsigned)(span >> 9))) : 0;
    const int nvl = (S - (nch - 1) * kChunkDraws) / VEC;  // qfull / qrem of the last chunk (unused by LOO mode)
    wave_sync();
    if constexpr (SPLIT) {
      // split pass: selection here, fit / smoothing / outputs in fit_rows_kernel (16 lanes per observation, pla_fit.h).
      // The second sum is brought to the true shift as well: e^-x = e^-x' e^(m - m')  (R < 690 keeps it finite)
      s2 *= exp_tab(delta, tb.tab);
      wave_select_split<SM, TB, (SM::Caps::kMaxTail + 63) / 64>(F, sm, tb, r, lane, M, m, mn, s1, s2, ncand, k1, sh, magic, c256, slow,
                                                                CandInLds<SM>{sm});
    } else {
      wave_back<T, VEC, LW, SM, TB, LW>(P, sm, tb, r, v, lane, S, M, mestM, logS, dbgs, m, mn, R, delta, s1, s2, ncand, k1,
                                               sh, magic, c256, nvl / kWave, nvl % kWave, slow, khat, loo, lppd);
    }
  }
  if constexpr (LWF) {
    // ---- weights mode: the row passes through the registers once more (its first chunk came in behind the last sweep),
    // each vector stored as lw = (raw - m) - L and replaced by the same vector of the following chunk; then the smoothed tail
    if (!slow) {
      const double L = loo;
      T* orow = reinterpret_cast<T*>(P.lw_out) + r * (int64_t)S;
#pragma unroll 1
      for (int ch = 0; ch < nch; ++ch) {
        const int Sc = (S - ch * kChunkDraws < kChunkDraws) ? S - ch * kChunkDraws : kChunkDraws;
        const int left = S - (ch + 1) * kChunkDraws;  // draws after this chunk
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(orow + (int64_t)ch * kChunkDraws, 0, Sc * (int)sizeof(T), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(row + (int64_t)(ch + 1 < nch ? ch + 1 : 0) * kChunkDraws), 0,
            left > 0 ? (left < kChunkDraws ? left : kChunkDraws) * (int)sizeof(T) : 0, 0x00020000);
        lw_store_chunk<T, VEC, true>(v, ro, rs_next, lane, (Sc / VEC) / kWave, m_run, L);
      }
      const int ntail = (int)lppd;
      if (ntail > 0) lw_patch_tail<T>(sm, tb, orow, lane, ntail, L);
    }
    // the next row's first chunk (weights mode never streams it behind the sweep)
    if (rp_next) issue_row_loads<T, VEC, 0>(v, rp_next, S < kChunkDraws ? S : kChunkDraws);
  }
  if (lane == 0) {
    if (slow && want.row == r) {
      // (comes round again: neither listed for the general kernel nor handed to the fit kernel -- a tail length of -1 in the
      // hand-over until the second attempt overwrites it)
      if constexpr (SPLIT) F.ws_s[r * F.ws_sstride + 5] = -1.0;
    } else if (slow) {
      const unsigned long long idx = atomicAdd(&F.counters[0], 1ull);
      F.slow_list[idx] = (unsigned)r + F.slow_base;
      if constexpr (SPLIT) F.ws_s[r * F.ws_sstride + 5] = -1.0;  // tail length -1: on the list, nothing for the fit kernel
    } else if constexpr (!SPLIT) {
      double *pd = P.diag, *pl = P.loo_i, *pp = P.lppd_i;
      asm volatile("" : "+s"(pd), "+s"(pl), "+s"(pp));  // (null tests inside the row loop, not hoisted into scalar registers it lacks)
      if (pd) pd[r] = khat;
      if constexpr (!LW) {
        if (pl) pl[r] = P.scale_value * loo;
        if (pp) pp[r] = lppd;
      }
    }
  }
}

template <typename T, int VEC, class CAP, bool SPLIT = false, bool LW = false>
__global__ __launch_bounds__(kWave * CAP::kWaves, 1) void wave_loo_chunked_kernel(RowsParams P, FastParams F) {
  using SM = std::conditional_t<LW && !SPLIT, WaveSmemLWT<CAP>, WaveSmemT<CAP>>;
  using TB = std::conditional_t<SPLIT, WaveTabOnly, WaveTablesT<CAP>>;
  constexpr int kWavesPerBlock = CAP::kWaves;
  __shared__ __attribute__((aligned(16))) SM scratch[kWavesPerBlock];
  __shared__ __attribute__((aligned(16))) TB tb;
  const int tid = threadIdx.x;
  for (int j = tid; j < kTabN; j += kWave * kWavesPerBlock) exp_table_entry(tb.tab, j);
  if constexpr (!SPLIT) {
    for (int j = tid; j < kLogTabN; j += kWave * kWavesPerBlock) log_table_entry(tb.lt, j);
    for (int j = tid; j < P.tail_count; j += kWave * kWavesPerBlock) tb.l1[j] = F.l1_table[j];
    if (tid < kWave) tb.bg[tid] = F.b_grid[tid];
  }
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
  SM& sm = scratch[wv];
  T v[kWaveSlots];
  const T* base = reinterpret_cast<const T*>(P.in);
  const int64_t w0 = (int64_t)blockIdx.x * kWavesPerBlock + wv, nw = (int64_t)gridDim.x * kWavesPerBlock;
  const int first = P.n_draws < kChunkDraws ? P.n_draws : kChunkDraws;
  const T* cur = w0 < P.n_obs ? base + PLA_ROW_OFFSET(P, w0) : nullptr;
  if (cur) issue_row_loads<T, VEC>(v, cur, first);
  // the rows of this wave in order; a row that wants a second attempt is scheduled right behind the row that is already
  // streaming in (that row's last sweep then streams the returning row's first chunk: ChunkRetry)
  int64_t r = w0, fresh = w0 + nw;  // the row in hand, the next row not yet started
  const ChunkRetry none{0.0, 0.0, -1, 0.0, 0.0, 0.0, 0};
  ChunkRetry now = none, pending = none;
  while (cur) {
    // what follows the row in hand: a returning row first, else the next fresh one
    const ChunkRetry after = pending.row >= 0 ? pending : none;
    const int64_t rn = after.row >= 0 ? after.row : fresh;
    if (after.row < 0) fresh += nw;
    pending.row = -1;
    const T* nxt = rn < P.n_obs ? base + PLA_ROW_OFFSET(P, rn) : nullptr;
    ChunkRetry want = none;
    wave_loo_row_chunked<T, VEC, SM, TB, SPLIT, LW>(P, F, sm, tb, r, v, cur, nxt, now, want);
    if (want.row >= 0) {
      if (nxt) {
        pending = want;  // behind the row that is streaming in
      } else {
        // (nothing is streaming in: this was the wave's last row.  It comes round at once -- its first chunk is requested
        // here, the one other place of issue, which only the last row of a wave reaches)
        issue_row_loads<T, VEC>(v, cur, first);
        now = want;
        continue;
      }
    }
    cur = nxt;
    r = rn;
    now = after;
  }
}

}  // namespace pla
