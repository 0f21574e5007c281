/*
 * pyloo_amd.h -- C ABI of the MI355X-native PSIS-LOO engine (libpyloo_amd.so).
 *
 * This is the drop-in boundary for the ONE hot path of jordandeklerk/pyloo that this
 * project accelerates.  The reference has no FFI of its own: the seam
 * is the Python call `wrap_xarray_ufunc(_psislw, ...)` that loops a 1-D NumPy routine over
 * observations.  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns an int status (0 = PLA_OK,
 *     negative = error) and never throws or aborts; pla_last_error() gives the text.
 *   - the caller owns every buffer; inputs are const and never modified (the reference
 *     deep-copies its input: psis.py:78, base.py:112); nothing is retained after return.
 *   - numeric trouble is in-band exactly as in the reference: k = +inf when the tail has
 *     <= 4 draws (psis.py:142-144), NaN weights for NaN rows, etc.  No status code for it.
 *   - `mem_space` says where ALL data pointers of that call live: PLA_HOST (the library
 *     stages through its own device buffers) or PLA_DEVICE (pointers are HIP device
 *     pointers on the engine's device, work is enqueued on `stream`, no host sync).
 *   - matrices are (n_obs, n_draws) with element strides (stride_obs, stride_draw) in
 *     ELEMENTS.  Two layouts are fast: stride_draw == 1 (draws contiguous), and stride_obs == 1 with
 *     stride_draw >= n_obs (observations contiguous: pyloo's stacked `(*obs, __sample__)` view of an ArviZ
 *     (chain, draw, *obs) array, loo.py:189) -- pla_psis_loo, pla_waic and (device pointers) pla_importance_weights
 *     transpose the latter block by block on the device.  Anything else runs on the strided general kernel (device pointers) or is refused (host pointers).
 *   - threads and streams: an engine is bound to one device and owns ONE workspace.  Every entry point that takes an
 *     engine holds the engine's mutex for the whole call, so concurrent calls from several threads are safe (they run one
 *     after the other); different engines are independent (no hidden global state).  PLA_DEVICE calls only ENQUEUE work
 *     that uses the workspace; the engine orders that work ACROSS STREAMS itself: every call records an event behind what
 *     it enqueued, and a call on another stream first makes its stream wait for the previous call's event -- two streams
 *     never run passes of one engine side by side, whatever the caller does (calls on one stream are ordered anyway).
 *     For passes that should overlap use one engine per stream.  Inside a stream capture no event is waited for or
 *     recorded: order captured work yourself.  Growing the workspace frees the old buffers with hipFree, which waits for
 *     the device, so launches already enqueued are never left with dangling pointers.
 *   - HIP graphs: a captured PLA_DEVICE call holds raw workspace pointers.  Size the workspace first (one eager call
 *     of the largest shape and tail count to be replayed), then pla_engine_set_frozen(eng, 1): from then on a call that
 *     would have to reallocate returns PLA_ERR_FROZEN instead of invalidating the graph.  The per-tail-count quantile
 *     tables are immutable once created, so eager calls with other tail counts do not disturb a captured graph.
 */
#ifndef PYLOO_AMD_H
#define PYLOO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLA_ABI_VERSION 4

/* status codes */
#define PLA_OK 0
#define PLA_ERR_ARG (-1)         /* bad argument (null pointer, n_draws < 2, bad enum ...) */
#define PLA_ERR_HIP (-2)         /* a HIP runtime call failed (text in pla_last_error) */
#define PLA_ERR_NOMEM (-3)       /* device or host allocation failed */
#define PLA_ERR_UNSUPPORTED (-4) /* shape outside what the kernels support */
#define PLA_ERR_NODEVICE (-5)    /* no usable AMD GPU */
#define PLA_ERR_FROZEN (-6)      /* the call needs to (re)allocate engine workspace while the engine is frozen */

/* element type of the log-likelihood / log-weight matrix */
#define PLA_F64 0
#define PLA_F32 1 /* read as f32, all arithmetic in f64 (hard part 5) */

#define PLA_HOST 0
#define PLA_DEVICE 1

/* importance-sampling method: ISMethod of base.py:18-23 */
#define PLA_PSIS 0
#define PLA_SIS 1
#define PLA_TIS 2

/* slots of the aggregate vector written by pla_psis_loo (all double) */
#define PLA_AGG_N 0           /* number of observations reduced                          */
#define PLA_AGG_SUM_LOO 1     /* sum_i loo_i                  (elpd_loo,  loo.py:326)      */
#define PLA_AGG_M2_LOO 2      /* sum_i (loo_i - mean)^2       (se, p_loo_se: loo.py:327,340) */
#define PLA_AGG_SUM_LPPD 3    /* sum_i lppd_i                 (loo.py:329-337)            */
#define PLA_AGG_N_HIGH 4      /* #(khat > good_k)             (loo.py:292-293)            */
#define PLA_AGG_N_NONFINITE 5 /* #(diagnostic not finite)                                 */
#define PLA_AGG_MIN_DIAG 6    /* min_i diagnostic             (min ESS, loo.py:306)       */
#define PLA_AGG_N_SLOW 7      /* rows that left the fast selection path (perf diagnostics) */
#define PLA_AGG_COUNT 8

typedef struct pla_engine pla_engine;

/* library / device ------------------------------------------------------------------- */
int pla_abi_version(void);
const char *pla_last_error(void); /* thread-local, valid until the next failing call */
int pla_device_count(int *count);

/* One engine per (process, GPU): owns the small device workspace (no per-call hipMalloc on
 * the PLA_DEVICE path, so calls can be captured in a hipGraph). */
int pla_engine_create(int device, pla_engine **out);
int pla_engine_destroy(pla_engine *eng);

/* frozen != 0: the workspace may no longer be reallocated (see "HIP graphs" above); 0 lifts it.  The reference has no
 * counterpart (it allocates per call: psis.py:92, base.py:125). */
int pla_engine_set_frozen(pla_engine *eng, int frozen);

/* M of base.py:139-141 / psis.py:89:  ceil(min(S/5, 3*sqrt(S/reff)));  cutoff_ind = -M-1 */
int pla_tail_count(int64_t n_draws, double reff, int64_t *tail_count);

/*
 * pla_psis_loo -- the fused per-observation LOO pass.
 * Replaces loo.py:286-342: compute_importance_weights(-ll) -> `log_weights += ll` ->
 * loo_i = scale * LSE_s(lw + ll) -> lppd_i = LSE_s(ll) - log S -> sums / variance / k counts,
 * i.e. the three Python loops of utils.py:137-142,171-175 over psis.py:114-160 and
 * utils.py:305-359, without ever materialising the (n_obs, n_draws) weight matrix.
 *
 *   ll          (n_obs, n_draws) log-likelihood, dtype PLA_F64 / PLA_F32, const
 *   method      PLA_PSIS / PLA_SIS / PLA_TIS
 *   tail_count  M from pla_tail_count (PSIS only; ignored otherwise)
 *   scale_value 1, -1 or -2 (loo.py:195-200); loo_i and the aggregates carry it
 *   good_k      threshold for PLA_AGG_N_HIGH (loo.py:249)
 *   diag        [n_obs] khat (PSIS) or ESS (SIS/TIS), always double (base.py:125); may be NULL
 *   loo_i       [n_obs] may be NULL          lppd_i [n_obs] may be NULL
 *   agg         [PLA_AGG_COUNT] may be NULL
 */
int pla_psis_loo(pla_engine *eng, const void *ll, int dtype, int64_t n_obs, int64_t n_draws,
                 int64_t stride_obs, int64_t stride_draw, int method, int64_t tail_count,
                 double scale_value, double good_k, int mem_space, void *stream, double *diag,
                 double *loo_i, double *lppd_i, double *agg);

/*
 * pla_importance_weights -- smoothed, truncated, normalised log weights AND the diagnostic.
 * Replaces the batched dispatch of base.py:160-166 / psis.py:100-106 (the `_multi_ufunc`
 * loop over `_psislw` / `_sislw` / `_tislw`).
 *
 *   logw   (n_obs, n_draws) log importance ratios (for LOO: -log_likelihood), const
 *   lw_out (n_obs, n_draws) C-contiguous, same dtype as logw (base.py:125 empty_like)
 *   diag   [n_obs] double
 */
int pla_importance_weights(pla_engine *eng, const void *logw, int dtype, int64_t n_obs,
                           int64_t n_draws, int64_t stride_obs, int64_t stride_draw, int method,
                           int64_t tail_count, int mem_space, void *stream, void *lw_out,
                           double *diag);

/*
 * pla_reduce_pointwise -- only the reductions of loo.py:326-342,292-293 over pointwise
 * vectors already on the device/host (used after sharded runs and by tests).
 */
int pla_reduce_pointwise(pla_engine *eng, const double *diag, const double *loo_i,
                         const double *lppd_i, int64_t n_obs, double good_k, int mem_space,
                         void *stream, double *agg);

/*
 * pla_waic -- the WAIC pass: one read of the matrix.
 * Replaces waic.py:109-160: NaN -> -1e10 and +-inf -> +-1e10 on load (112-135),
 * lppd_i = LSE_s(ll) - log S (137-143, utils.py:305-359), var_i = population variance over draws (145),
 * waic_i = scale * (lppd_i - var_i) (158) and the sums of 159-161.
 *
 *   lppd_i, var_i, waic_i   [n_obs] double, each may be NULL
 *   agg  [PLA_AGG_COUNT] may be NULL, slots reused as
 *        PLA_AGG_N          n
 *        PLA_AGG_SUM_LOO    sum_i waic_i            (elpd_waic, waic.py:160)
 *        PLA_AGG_M2_LOO     sum_i (waic_i - mean)^2 (se = sqrt(M2), waic.py:159)
 *        PLA_AGG_SUM_LPPD   sum_i var_i             (p_waic, waic.py:161)
 *        PLA_AGG_N_HIGH     #(var_i > 0.4)          (warning, waic.py:147)
 *        PLA_AGG_MIN_DIAG   min_i var_i
 *        PLA_AGG_N_SLOW     entries replaced on load (NaN or +-inf: the front's two warnings)
 */
int pla_waic(pla_engine *eng, const void *ll, int dtype, int64_t n_obs, int64_t n_draws,
             int64_t stride_obs, int64_t stride_draw, double scale_value, int mem_space, void *stream,
             double *lppd_i, double *var_i, double *waic_i, double *agg);

/*
 * pla_psis_loo_rows / pla_waic_rows -- the same two passes over a SELECTION of the rows of a resident matrix:
 * the subsampled LOO of loo_subsample.py:316-330 (`log_likelihood.isel(...)` on the sampled observations, then
 * compute_importance_weights + logsumexp, 373-386, and the variance over draws of the sampled rows, 389) without
 * materialising the gathered copy.
 *
 *   ll, n_obs, n_draws, strides   the full matrix, as in pla_psis_loo
 *   row_index  [n_rows] int64 observation indices into the matrix (repeats allowed), in the memory space of `ll`
 *              (device pointer for PLA_DEVICE).  Host lists are range-checked (PLA_ERR_ARG); device lists are
 *              clamped into [0, n_obs) on the device before use -- validate them before uploading.
 *   outputs    compact: entry r belongs to observation row_index[r]; [n_rows] each; agg as in the full-matrix call
 */
int pla_psis_loo_rows(pla_engine *eng, const void *ll, int dtype, int64_t n_obs, int64_t n_draws,
                      int64_t stride_obs, int64_t stride_draw, const int64_t *row_index, int64_t n_rows,
                      int method, int64_t tail_count, double scale_value, double good_k, int mem_space,
                      void *stream, double *diag, double *loo_i, double *lppd_i, double *agg);
int pla_waic_rows(pla_engine *eng, const void *ll, int dtype, int64_t n_obs, int64_t n_draws,
                  int64_t stride_obs, int64_t stride_draw, const int64_t *row_index, int64_t n_rows,
                  double scale_value, int mem_space, void *stream, double *lppd_i, double *var_i,
                  double *waic_i, double *agg);

/*
 * pla_e_loo -- PSIS-weighted expectations of a same-shape matrix and their function-specific Pareto k.
 * Replaces, per observation, e_loo.py:214-236: `_normalize_log_weights` + `_compute_weighted_mean` (430-437, 557-559),
 * `_compute_weighted_variance` / `_wvar_func` (440-459, 518-531; sd = sqrt(variance), 462-465) and `compute_pareto_k` ->
 * `k_hat` (266-390) for h = x (mean), h = x^2 (variance / sd) and h = None (quantiles), i.e. the `wrap_xarray_ufunc` loops of
 * e_loo.py:315-324, 448-457 and the callers on top of psislw: loo_score.py:227,312, loo_predictive_metric.py:208.
 *
 *   x            (n_obs, n_draws) draws to average (posterior-predictive or posterior values), const
 *   log_weights  (n_obs, n_draws) log importance weights, any normalisation (the smoothed weights of
 *                pla_importance_weights; `weights=` callers pass log(weights), e_loo.py:202-203)
 *   log_ratios   (n_obs, n_draws) raw log ratios for the diagnostics, or NULL = log_weights (e_loo.py:223-224)
 *                -- all three share dtype, shape and strides
 *   tail_len     draws per tail in k_hat (20: e_loo.py:269); >= 5
 *   mean, variance, k_mean, k_var, k_ratio   [n_obs] double, each may be NULL
 *                k_mean = k_hat(x, lr), k_var = k_hat(x^2, lr), k_ratio = k_hat(None, lr)
 * In-band semantics as in the reference: NaN / inf in x flow into mean and variance by IEEE rules and switch k to the
 * ratio-only value (e_loo.py:359-366); constant x gives variance 0 (520-521).  k_hat is reproduced AS THE REFERENCE EVALUATES
 * IT, including the descending tails it hands to `_gpdfit` (see csrc/pla_eloo.h for what that implies).
 */
int pla_e_loo(pla_engine *eng, const void *x, const void *log_weights, const void *log_ratios, int dtype,
              int64_t n_obs, int64_t n_draws, int64_t stride_obs, int64_t stride_draw, int64_t tail_len,
              int mem_space, void *stream, double *mean, double *variance, double *k_mean, double *k_var,
              double *k_ratio);

/*
 * pla_e_loo_quantiles -- PSIS-weighted quantiles of the draws (e_loo(type="quantile")).
 * Replaces `_compute_weighted_quantiles` (e_loo.py:468-515: the `np.ndindex` loop over observations and probabilities) and
 * `_weighted_quantile` (534-554): argsort + cumulative normalised weights + linear interpolation between the two draws that
 * bracket `prob`; np.quantile(x, prob) when the weights are all close (536-537).  No sort on the device: a weighted radix
 * selection per (observation, prob) -- see csrc/pla_eloo.h.
 *
 *   x, log_weights  (n_obs, n_draws), same dtype and strides, as in pla_e_loo
 *   probs           [n_probs] HOST array (whatever mem_space says), each strictly between 0 and 1 (e_loo.py:158-159)
 *   out             [n_obs][n_probs] double, in the memory space of the matrices
 * The Pareto k of this type is pla_e_loo's k_ratio (e_loo.py:229-230).
 */
int pla_e_loo_quantiles(pla_engine *eng, const void *x, const void *log_weights, int dtype, int64_t n_obs,
                        int64_t n_draws, int64_t stride_obs, int64_t stride_draw, const double *probs,
                        int64_t n_probs, int mem_space, void *stream, double *out);

/* Timing of the dominant kernel, measured with hipEvents on the launch stream.
 * enable != 0 brackets every main-kernel launch with events; pla_engine_kernel_ms returns the
 * accumulated milliseconds and launch count since the last call (it synchronises the events). */
int pla_engine_set_timing(pla_engine *eng, int enable);
int pla_engine_kernel_ms(pla_engine *eng, double *total_ms, int64_t *launches);
/* Same, for the first (dominant) kernel alone of the passes that ran as two kernels (the split PSIS-LOO pass: the
 * wave kernel up to the tail selection); device-pointer calls only.  Read it before or after pla_engine_kernel_ms. */
int pla_engine_first_kernel_ms(pla_engine *eng, double *total_ms, int64_t *launches);

/* Which kernels the engine's last PSIS-LOO / weights call launched, as text ("wave_loo_kernel<double> (streamed) +
 * fit_rows_stream_kernel beside it + ..."): for benchmark records, so that what a roofline line names is what ran.
 * Copies at most cap - 1 characters and a terminating 0 into buf. */
int pla_engine_last_kernels(pla_engine *eng, char *buf, int cap);

/* Run-time switches of the library that are SET in this process's environment, as "NAME=value NAME=value" ("" when none is):
 * for benchmark records.  The shipped library reads PLA_PIPE, PLA_STREAM_PATIENCE_US, PLA_FORCE_PATH, PLA_INGEST_TRANSPOSE and
 * PLA_INGEST_BLOCK_MB -- path selectors, every setting of which computes the same results (csrc/pla_launch.h); builds with
 * -DPLA_EXPERIMENT read more (ablation, grids, priorities), and the text then starts with "EXPERIMENT-BUILD".
 * Copies at most cap - 1 characters and a terminating 0 into buf. */
int pla_env_overrides(char *buf, int cap);

/* Streamed PSIS-LOO passes (the fit kernel running beside the sweep): *gave_up = how many passes since the last call the fit
 * kernel stopped waiting for the sweep (the two were not run side by side: a serialising profiler, a co-tenant holding the
 * CUs) and left the rest to the plain fit kernel behind it -- correct, but the pass then costs up to 20 ms more, which a
 * benchmark record should show.  Synchronises the device. */
int pla_engine_stream_stats(pla_engine *eng, int64_t *gave_up);

/*
 * Observation-sharded runs (loo.py:326-342 across devices): every device reduces its own block of
 * observations to one aggregate vector; the vectors are exchanged with ONE all-reduce (sum) of a world x PLA_AGG_COUNT
 * table in which every rank has filled its own row -- the host's collective library does that (torch.distributed / RCCL
 * in pyloo_amd.sharded; any all-reduce of doubles will do) -- and merged with the pairwise update of Chan, Golub & LeVeque.
 * Device pointers, the caller's stream, one small kernel each:
 *   pla_aggregate_pack   table[world][PLA_AGG_COUNT] = 0 except row `rank` = agg   (before the all-reduce)
 *   pla_aggregate_merge  out[PLA_AGG_COUNT] = the merged aggregates of the table   (after it; identical on every rank)
 */
int pla_aggregate_pack(pla_engine *eng, const double *agg, int rank, int world, double *table, void *stream);
int pla_aggregate_merge(pla_engine *eng, const double *table, int world, double *out, void *stream);

/* Synthetic benchmark input, generated on the device:
 *   u = splitmix64(seed ^ (i*S + s)) -> 53-bit uniform in (0,1) -> E = -log1p(-u)
 *   ll[i,s] = -k_i*E + c_i,  c_i = -1 - (i mod 7)/4,  k_i = k_lo + (k_hi-k_lo)*U(splitmix64(~seed ^ i))
 *   rows with (i mod 10) in {0,3,6} draw k_i from [heavy_lo, heavy_hi) when heavy_hi > heavy_lo.
 * row0 offsets the observation index so shards of one matrix can be generated per rank. */
int pla_fill_synthetic(pla_engine *eng, void *ll_device, int dtype, int64_t n_obs, int64_t n_draws,
                       int64_t row0, uint64_t seed, double k_lo, double k_hi, double heavy_lo,
                       double heavy_hi, void *stream);

/* The same marginals as MCMC delivers them (bench.py --rows chain_ar1): `chains` chains stacked chain-major along the draws
 * (the (chain, draw) -> __sample__ stack of loo.py:189), every chain a stationary AR(1) sequence in the draw index with
 * coefficient rho (Gaussian copula, Exp(1) marginals), k_i ~ U(k_lo, k_hi) and c_i as above, plus an offset ~ N(0, offset_sd^2)
 * of each chain's log-likelihoods. */
int pla_fill_synthetic_chains(pla_engine *eng, void *ll_device, int dtype, int64_t n_obs, int64_t n_draws,
                              int64_t row0, uint64_t seed, int chains, double rho, double offset_sd,
                              double k_lo, double k_hi, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PYLOO_AMD_H */
