"""``loo_subsample()`` -- PSIS-LOO on a subsample of the observations (pyloo/loo_subsample.py:37-607) with the
device doing every pass over the matrix.

What runs where:

* approximation of ``loo_i`` for ALL observations (``approximations/lpd.py:51-65``,
  ``approximations/importance_sampling.py:60-73``): one streaming pass of the SIS / TIS kernel (``pla_psis_loo``),
  which yields ``log mean_s exp(ll)`` ("lpd") and the importance-sampling ``loo_i`` ("sis", "tis") together;
* drawing the subsample (``estimators/base.py:75-122``): host, the same ``np.random`` calls as the reference, so a
  seeded run selects the same observations;
* PSIS on the sampled rows (loo_subsample.py:316-386) and their variance over draws (389): ``pla_psis_loo_rows`` /
  ``pla_waic_rows`` read the selected rows of the resident matrix in place -- no gathered copy;
* the survey-sampling estimators (``estimators/difference.py:60-112``, ``hansen_hurwitz.py:61-91``,
  ``srs.py:62-84``): a few vector operations on ``m`` numbers, host.

* the ``log_p`` / ``log_q`` posterior correction (loo_subsample.py:333-370): ``importance_resample`` smooths the ratios on
  the device and draws with the reference's generator calls; the sampled rows are gathered with their draws re-drawn (a
  copy of ``m`` rows) and go through the same two passes.

Not carried over: ``loo_approximation="plpd"`` with a user log-likelihood function (needs the model; without one the
reference falls back to the mean log-likelihood, which is what "plpd" does here, with the same warning) and
``update_subsample``.
"""

import warnings
from collections import namedtuple

import numpy as np

from .base import ISMethod, tail_count_for
from .elpd import ELPDData
from .engine import _is_torch_tensor, get_engine
from .loo import _relative_efficiency, _replace_nan, _scale_value, loo
from .rcparams import rcParams
from .utils import get_log_likelihood, stack_samples, to_inference_data, wrap_obs

__all__ = ["loo_subsample", "loo_subsample_from_matrix", "subsample_indices", "SubsampleIndices", "Estimate",
           "srs_estimate", "diff_srs_estimate", "hansen_hurwitz_estimate", "compute_sampling_probabilities"]

APPROXIMATIONS = ("plpd", "lpd", "tis", "sis")   # constants.py LooApproximationMethod
ESTIMATORS = ("diff_srs", "hh_pps", "srs")       # constants.py EstimatorMethod

SubsampleIndices = namedtuple("SubsampleIndices", ["idx", "m_i"])
# y_hat: point estimate of the population total; v_y_hat: its subsampling variance; hat_v_y: total variance
Estimate = namedtuple("Estimate", ["y_hat", "v_y_hat", "hat_v_y", "m", "N", "subsampling_SE"])


# ----------------------------------------------------------------------------------------- sampling
def compute_sampling_probabilities(elpd_loo_approximation):
    """Selection probabilities proportional to |approximation| (hansen_hurwitz.py:113-125)."""
    size = np.abs(np.asarray(elpd_loo_approximation, dtype=float))
    if np.all(size <= 0):
        size = np.ones_like(size)
    size = np.maximum(size, np.finfo(float).tiny)
    return size / size.sum()


def subsample_indices(estimator, elpd_loo_approximation, observations):
    """Draw the subsample (estimators/base.py:75-122); uses the global ``np.random`` state like the reference."""
    n = len(elpd_loo_approximation)
    if estimator == "hh_pps":  # with replacement, probability proportional to size
        size = np.abs(elpd_loo_approximation)
        picks = np.random.choice(n, size=observations, replace=True, p=size / size.sum())
        idx, counts = np.unique(picks, return_counts=True)
        return SubsampleIndices(idx=idx, m_i=counts)
    if estimator in ("diff_srs", "srs"):  # simple random sampling without replacement
        if observations > n:
            raise ValueError("Number of observations cannot exceed total sample size when using SRS without replacement")
        idx = np.sort(np.random.choice(n, size=observations, replace=False))
        return SubsampleIndices(idx=idx, m_i=np.ones_like(idx))
    raise ValueError(f"Unknown estimator: {estimator}")


# ----------------------------------------------------------------------------------------- estimators
def srs_estimate(y, N):
    """Expansion estimator under simple random sampling (srs.py:62-84)."""
    y = np.asarray(y, dtype=float)
    N = int(N)
    m = len(y)
    s2 = np.var(y, ddof=1)
    v_sub = N ** 2 * (1 - m / N) * s2 / m
    return Estimate(N * np.mean(y), v_sub, N * s2, m, N, np.sqrt(v_sub))


def diff_srs_estimate(y_approx, y, y_idx):
    """Difference estimator (difference.py:60-112; Magnusson et al. 2020): the total of the approximations plus the
    expanded mean difference on the sample."""
    y_approx = np.asarray(y_approx, dtype=float)
    y = np.asarray(y, dtype=float)
    y_idx = np.asarray(y_idx)
    if len(y) != len(y_idx):
        raise ValueError("y and y_idx must have same length")
    if np.max(y_idx) >= len(y_approx):
        raise ValueError("y_idx contains invalid indices")
    N, m = len(y_approx), len(y)
    approx_m = y_approx[y_idx]
    diff = y - approx_m
    total_approx = np.sum(y_approx)
    total_approx_sq = np.sum(y_approx ** 2)
    t_e = N * np.mean(diff)
    t_eps = N * np.mean(y ** 2 - approx_m ** 2)
    y_hat = total_approx + t_e
    if m > 1:
        v_sub = N ** 2 * (1 - m / N) * np.var(diff, ddof=1) / m
        v_tot = (total_approx_sq + t_eps) - (t_e ** 2 - v_sub + 2 * total_approx * y_hat - total_approx ** 2) / N
    else:
        v_sub = v_tot = np.inf
    return Estimate(y_hat, v_sub, v_tot, m, N, np.sqrt(v_sub))


def hansen_hurwitz_estimate(z, m_i, y, N):
    """Hansen-Hurwitz estimator for sampling with replacement, probability proportional to size
    (hansen_hurwitz.py:61-91; Magnusson et al. 2019)."""
    z = np.asarray(z, dtype=float)
    m_i = np.asarray(m_i)
    y = np.asarray(y, dtype=float)
    N = int(N)
    if not np.all(z > 0):
        raise ValueError("All probabilities (z) must be positive")
    if not np.all(m_i > 0):
        raise ValueError("All sample counts (m_i) must be positive")
    if not len(z) == len(m_i) == len(y):
        raise ValueError("All input arrays must have same length")
    z = z / z.sum()
    m = m_i.sum()
    ratio = y / z
    y_hat = np.sum(m_i * ratio) / m
    v_sub = (np.sum(m_i * (ratio - y_hat) ** 2) / m) / (m - 1)
    v_tot = np.sum(m_i * (y ** 2 / z)) / m + v_sub / N - y_hat ** 2 / N
    return Estimate(y_hat, v_sub, v_tot, m, N, np.sqrt(v_sub))


# ----------------------------------------------------------------------------------------- engine passes
def _to_host(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def _thin(matrix, n_draws):
    """approximations/base.py:60-102: evenly spaced draws for the approximation pass."""
    if n_draws is None:
        return matrix
    n_samples = matrix.shape[-1]
    if n_draws > n_samples:
        raise ValueError(f"Requested {n_draws} draws but only {n_samples} are available")
    cols = np.linspace(0, n_samples - 1, n_draws, dtype=int)
    if _is_torch_tensor(matrix):
        import torch

        return matrix.index_select(1, torch.as_tensor(cols, device=matrix.device)).contiguous()
    return np.ascontiguousarray(matrix[:, cols])


def _approximation(eng, matrix, kind, n_draws):
    """``loo_i`` approximation for every observation: one pass of the streaming SIS / TIS kernel."""
    if kind == "plpd":
        # approximations/plpd.py:84-95 without a model: the mean log-likelihood, flagged like the reference does
        warnings.warn(
            "Using approximate PLPD calculation. For better accuracy, provide "
            "log likelihood and data to compute log likelihoods directly.",
            UserWarning,
            stacklevel=3,
        )
        m = _thin(matrix, n_draws)
        return _to_host(m.mean(dim=1)) if _is_torch_tensor(m) else np.asarray(m, dtype=np.float64).mean(axis=1)
    res = eng.psis_loo(_thin(matrix, n_draws), 0, "tis" if kind == "tis" else "sis", 1.0, 0.7, aggregate=False)
    return _to_host(res["lppd_i"] if kind == "lpd" else res["loo_i"]).astype(np.float64)


def importance_resample(log_p, log_q, method="psis", seed=None):
    """Indices of the draws after importance resampling from an approximate posterior
    (``loo_approximate_posterior.importance_resample``, loo_approximate_posterior.py:437-536, as ``loo_subsample`` calls it at
    loo_subsample.py:341-346): the log ratios ``log_p - log_q`` are Pareto-smoothed (``"psis"``: then drawn WITHOUT
    replacement, i.e. a weighted permutation of the draws; ``"psir"``: with replacement) or only normalised (``"sis"``), and
    ``numpy.random.RandomState(seed).choice`` draws as many indices as there are draws.  The smoothing runs on the device
    (``pla_importance_weights``); the generator, its fall-backs and the warnings are the reference's.  Non-finite ratios end as
    they do there -- in the exception numpy raises when the map back to the original positions runs out of bounds -- which
    ``loo_subsample`` reports as its "Importance resampling failed" warning."""
    log_p = np.asarray(log_p, dtype=np.float64)
    log_q = np.asarray(log_q, dtype=np.float64)
    rng = np.random.RandomState(seed) if seed is not None else np.random.RandomState()
    draws = len(log_p)
    logiw = log_p - log_q
    valid_idx = np.isfinite(logiw)
    if not np.all(valid_idx):
        warnings.warn(f"Found {np.sum(~valid_idx)} non-finite importance weights. These will be excluded.", UserWarning, stacklevel=2)
        if np.sum(valid_idx) == 0:
            raise ValueError("No valid importance weights found.")
        logiw = logiw[valid_idx]
    else:
        valid_idx = slice(None)
    replace = method == "psir"
    if method in ["psis", "psir"]:
        try:
            n = len(logiw)
            lw, _ = get_engine(None).importance_weights(np.ascontiguousarray(logiw.reshape(1, n)), tail_count_for(n, 1.0),
                                                        ISMethod.PSIS.value)
            logiw = np.asarray(lw, dtype=np.float64).reshape(n)
        except Exception as e:  # noqa: BLE001  (the reference's own catch-all around psislw)
            warnings.warn(f"PSIS smoothing failed: {str(e)}.", UserWarning, stacklevel=2)
    else:
        m = np.max(logiw)
        logiw = logiw - (m + np.log(np.sum(np.exp(logiw - m))))
    with np.errstate(over="ignore"):
        p = np.exp(logiw)
    p = p / np.sum(p)
    try:
        indices_subset = rng.choice(draws, size=draws, replace=replace, p=p)
    except ValueError as e:
        if "Fewer non-zero entries in p than size" in str(e) and not replace:
            warnings.warn("Not enough non-zero weights for sampling without replacement. Switching to sampling with replacement.",
                          UserWarning, stacklevel=2)
            indices_subset = rng.choice(draws, size=draws, replace=True, p=p)
        else:
            warnings.warn(f"Resampling failed: {str(e)}. Using random indices.", UserWarning, stacklevel=2)
            indices_subset = rng.choice(draws, size=draws)
    if isinstance(valid_idx, np.ndarray):
        return np.where(valid_idx)[0][indices_subset]
    return indices_subset


def loo_subsample_from_matrix(log_likelihood, observations=100, loo_approximation="lpd", estimator="diff_srs",
                              loo_approximation_draws=None, reff=1.0, scale=None, pointwise=False, draw_index=None):
    """Subsampled PSIS-LOO from an ``(n_obs, n_draws)`` matrix (NumPy array, or torch CUDA tensor: device-resident, the
    sampled rows are read in place).  ``draw_index``: the draws of the SAMPLED rows are taken in this order / multiplicity before
    PSIS-LOO and the variance over draws (the posterior correction, loo_subsample.py:348-356).
    Returns ``(ELPDData, SubsampleIndices, Estimate)``; see :func:`loo_subsample`."""
    kind = str(loo_approximation).lower()
    if kind not in APPROXIMATIONS:
        raise ValueError(f"Invalid loo_approximation '{loo_approximation}'. Must be one of: {', '.join(APPROXIMATIONS)}")
    est = str(estimator).lower()
    if est not in ESTIMATORS:
        raise ValueError(f"Invalid estimator '{estimator}'. Must be one of: {', '.join(ESTIMATORS)}")
    scale, scale_value = _scale_value(scale)
    n_data_points, n_samples = log_likelihood.shape
    if isinstance(observations, (int, np.integer)) and not isinstance(observations, bool):
        if observations <= 0 or observations > n_data_points:
            raise ValueError(f"Number of observations must be between 1 and {n_data_points}, got {observations}")
    elif isinstance(observations, np.ndarray):
        if not np.issubdtype(observations.dtype, np.integer):
            raise TypeError("observations array must contain integers")
        if observations.min() < 0 or observations.max() >= n_data_points:
            raise ValueError(
                f"Observation indices must be between 0 and {n_data_points - 1}, "
                f"got range [{observations.min()}, {observations.max()}]"
            )
    else:
        raise TypeError("observations must be None, an integer, or an array of integers")

    eng = get_engine(log_likelihood.device.index if _is_torch_tensor(log_likelihood) else None)
    approx = _approximation(eng, log_likelihood, kind, loo_approximation_draws)
    if isinstance(observations, np.ndarray):
        indices = SubsampleIndices(idx=observations, m_i=np.ones_like(observations))
    else:
        indices = subsample_indices(est, approx, int(observations))

    # PSIS-LOO and the variance over draws on the sampled rows only (loo_subsample.py:373-389)
    good_k = min(1 - 1 / np.log10(n_samples), 0.7)
    M = tail_count_for(n_samples, reff)
    if M + 1 > n_samples:
        raise IndexError(f"index {-M - 1} is out of bounds for axis 0 with size {n_samples}")
    if draw_index is not None and _is_torch_tensor(log_likelihood):  # the m sampled rows with their draws re-drawn: a small copy
        import torch

        rows_t = torch.as_tensor(np.asarray(indices.idx), device=log_likelihood.device, dtype=torch.long)
        draws_t = torch.as_tensor(np.asarray(draw_index), device=log_likelihood.device, dtype=torch.long)
        sub = log_likelihood.index_select(0, rows_t).index_select(1, draws_t).contiguous()
        res = eng.psis_loo(sub, M, ISMethod.PSIS.value, scale_value, good_k, aggregate=False)
        var_m = eng.waic(sub, 1.0, aggregate=False)["var_i"]
    elif draw_index is not None:
        sub = np.ascontiguousarray(np.asarray(log_likelihood)[indices.idx][:, np.asarray(draw_index)])
        res = eng.psis_loo(sub, M, ISMethod.PSIS.value, scale_value, good_k, aggregate=False)
        var_m = eng.waic(sub, 1.0, aggregate=False)["var_i"]
    elif _is_torch_tensor(log_likelihood):  # resident matrix: the sampled rows are read in place
        res = eng.psis_loo(log_likelihood, M, ISMethod.PSIS.value, scale_value, good_k, aggregate=False, rows=indices.idx)
        var_m = eng.waic(log_likelihood, 1.0, aggregate=False, rows=indices.idx)["var_i"]
    else:  # host matrix (possibly an observations-fastest view): only the m sampled rows travel
        sub = np.ascontiguousarray(np.asarray(log_likelihood)[indices.idx])
        res = eng.psis_loo(sub, M, ISMethod.PSIS.value, scale_value, good_k, aggregate=False)
        var_m = eng.waic(sub, 1.0, aggregate=False)["var_i"]
    loo_m = _to_host(res["loo_i"]).astype(np.float64)
    khat = _to_host(res["diag"]).astype(np.float64)
    p_loo_m = _to_host(var_m).astype(np.float64)

    if est == "hh_pps":
        z = compute_sampling_probabilities(approx)[indices.idx]
        estimates = hansen_hurwitz_estimate(z, indices.m_i, loo_m, n_data_points)
        p_est = hansen_hurwitz_estimate(z, indices.m_i, p_loo_m, n_data_points)
    elif est == "srs":
        estimates = srs_estimate(loo_m, n_data_points)
        p_est = srs_estimate(p_loo_m, n_data_points)
    else:
        estimates = diff_srs_estimate(approx, loo_m, indices.idx)
        p_est = srs_estimate(p_loo_m, n_data_points)

    se, sub_se = np.sqrt(estimates.hat_v_y), np.sqrt(estimates.v_y_hat)
    p_loo, p_loo_se, p_loo_sub_se = p_est.y_hat, np.sqrt(p_est.hat_v_y), np.sqrt(p_est.v_y_hat)
    looic, looic_se, looic_sub_se = -2 * estimates.y_hat, 2 * se, 2 * sub_se

    warn_mg = False
    if est == "srs":  # loo_subsample.py:451-461 (the branch reads the PSIS diagnostic as if it were an ESS)
        min_ess = np.min(khat)
        if min_ess < n_samples * 0.1:
            warnings.warn(
                f"Low effective sample size detected (minimum ESS: {min_ess:.1f}). This"
                " indicates that the importance sampling approximation may be"
                " unreliable. Consider using PSIS which is more robust to such cases.",
                UserWarning,
                stacklevel=3,
            )
            warn_mg = True
    else:
        max_k = np.nanmax(khat) if not np.all(np.isnan(khat)) else 0
        if max_k > good_k:
            warnings.warn(
                "Estimated shape parameter of Pareto distribution is greater than"
                f" {good_k:.2f} for {np.sum(khat > good_k)} observations. This indicates that"
                " importance sampling may be unreliable because the marginal posterior"
                " and LOO posterior are very different.",
                UserWarning,
                stacklevel=3,
            )
            warn_mg = True

    loo_full = np.full(n_data_points, np.nan)
    loo_full[indices.idx] = loo_m
    seen = loo_full[~np.isnan(loo_full)]
    if len(seen) > 0 and np.allclose(seen, seen[0]):
        warnings.warn(
            "The point-wise LOO is the same with the sum LOO, please double check "
            "the Observed RV in your model to make sure it returns element-wise logp.",
            UserWarning,
            stacklevel=3,
        )

    head = [("elpd_loo", estimates.y_hat), ("se", se), ("p_loo", p_loo), ("p_loo_se", p_loo_se),
            ("p_loo_subsampling_se", p_loo_sub_se), ("n_samples", n_samples), ("n_data_points", n_data_points),
            ("warning", warn_mg)]
    tail = [("scale", scale), ("good_k", good_k), ("subsampling_SE", sub_se), ("subsample_size", len(indices.idx)),
            ("looic", looic), ("looic_se", looic_se), ("looic_subsamp_se", looic_sub_se)]
    if pointwise:  # index order of loo_subsample.py:545-585
        rows = head + [("loo_i", loo_full)] + tail + [("pareto_k", khat), ("method", "loo_subsample")]
    else:          # loo_subsample.py:506-541
        rows = head + tail + [("method", "loo_subsample")]
    out = ELPDData(data=[v for _, v in rows], index=[k for k, _ in rows])
    out.method = "loo_subsample"
    return out, indices, estimates


def loo_subsample(data, observations=100, loo_approximation="plpd", estimator="diff_srs", loo_approximation_draws=None,
                  log_p=None, log_q=None, pointwise=None, var_name=None, reff=None, scale=None, resample_method="psis",
                  seed=None):
    """Approximate LOO-CV for large data by subsampling, with ``pyloo.loo_subsample``'s parameters, warnings and result
    layout (loo_subsample.py:37-607).  ``observations``: number of observations to draw, an integer index array, or
    ``None`` for the full ``loo()``."""
    idata = to_inference_data(data)
    log_likelihood = get_log_likelihood(idata, var_name=var_name)
    pointwise = rcParams["stats.ic_pointwise"] if pointwise is None else pointwise
    kind = str(loo_approximation).lower()
    if kind not in APPROXIMATIONS:
        raise ValueError(f"Invalid loo_approximation '{loo_approximation}'. Must be one of: {', '.join(APPROXIMATIONS)}")
    if str(estimator).lower() not in ESTIMATORS:
        raise ValueError(f"Invalid estimator '{estimator}'. Must be one of: {', '.join(ESTIMATORS)}")
    matrix, obs_shape, obs_dims, coords = stack_samples(log_likelihood)
    n_samples = matrix.shape[-1]
    scale, _ = _scale_value(scale)
    if reff is None:
        reff = _relative_efficiency(idata, n_samples)
    matrix = _replace_nan(matrix)
    if observations is None:  # loo_subsample.py:252-259
        return loo(data=data, pointwise=pointwise, var_name=var_name, reff=reff, scale=scale)
    draw_index = None
    if log_p is not None and log_q is not None:  # posterior correction: loo_subsample.py:333-370
        if len(log_p) != len(log_q):
            raise ValueError(f"log_p and log_q must have the same length, got {len(log_p)} and {len(log_q)}")
        try:
            draw_index = importance_resample(log_p=log_p, log_q=log_q, method=resample_method, seed=seed)
            if len(draw_index) != n_samples:
                # the reference reshapes the resampled draws to the stacked shape (loo_subsample.py:348-356): any other length
                # raises there and lands in the fallback below
                raise ValueError(f"cannot reshape array of size {len(draw_index)} into shape ({n_samples},)")
            if len(draw_index) and (np.min(draw_index) < 0 or np.max(draw_index) >= n_samples):
                raise IndexError(f"index {int(np.max(draw_index))} is out of bounds for axis 0 with size {n_samples}")
        except Exception as e:  # noqa: BLE001  (the reference's catch-all: loo_subsample.py:363-370)
            warnings.warn(f"Importance resampling failed: {str(e)}. Falling back to original samples.", UserWarning, stacklevel=2)
            draw_index = None
    out, indices, estimates = loo_subsample_from_matrix(matrix, observations, kind, estimator, loo_approximation_draws,
                                                        reff, scale, pointwise, draw_index=draw_index)
    if pointwise:
        out["loo_i"] = wrap_obs(np.asarray(out["loo_i"]), obs_shape, obs_dims, coords, "loo_i")
    # what update_subsample() of the reference reads back (loo_subsample.py:589-596)
    object.__setattr__(out, "estimates", estimates)
    object.__setattr__(out, "subsample_indices", indices)
    object.__setattr__(out, "loo_approximation", loo_approximation)
    object.__setattr__(out, "estimator", estimator)
    return out
