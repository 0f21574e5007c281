"""``loo_score()`` -- LOO-CRPS / LOO-SCRPS with the reference's signature and result object (pyloo/loo_score.py:19-274), on
the HIP engine.

Per observation the reference needs two PSIS-weighted expectations (loo_score.py:219-245, 277-323):

    E|X - X'|  over ``permutations`` random pairings of the draws: ``psislw`` of the JOINT log ratios
               ``-log_lik - log_lik[shuffle]`` and ``e_loo`` of ``|x - x2[shuffle]|`` under those weights;
    E|X - y|   ``psislw(-log_lik)`` and ``e_loo`` of ``|x - y|``;

then CRPS = E|X - X'| / 2 - E|X - y|, SCRPS = -E|X - y| / E|X - X'| - log(E|X - X'|) / 2 (326-346).  Each ``psislw`` /
``e_loo`` pair is ``pla_importance_weights`` + ``pla_e_loo`` on the stacked ``(n_obs, n_draws)`` matrices; the shuffle
is NumPy's global generator, exactly where the reference draws it (305), so a seeded run pairs the same draws.
``score_from_matrix`` takes the matrices directly (NumPy or torch CUDA tensors)."""

import warnings
from dataclasses import dataclass
from typing import Any

import numpy as np

from .base import tail_count_for
from .engine import _is_torch_tensor, get_engine
from .rcparams import rcParams
from .utils import get_log_likelihood, group_variable, stack_samples, to_inference_data, wrap_obs

__all__ = ["loo_score", "score_from_matrix", "LooScoreResult"]


@dataclass
class LooScoreResult:
    """loo_score.py:19-45."""

    estimates: np.ndarray
    pointwise: np.ndarray
    pareto_k: Any = None
    good_k: Any = None
    warning: Any = None


def _crps(exx, exy, scale=False):
    """loo_score.py:326-346."""
    if scale:
        return -exy / exx - 0.5 * np.log(exx)
    return 0.5 * exx - exy


def _weighted_mean(eng, values, ratios, reff):
    """``e_loo(values, log_weights=psislw(ratios)[0], log_ratios=ratios).value`` and the Pareto k of the weights."""
    lw, k = eng.importance_weights(ratios, tail_count_for(ratios.shape[-1], reff), "psis")
    return eng.e_loo(values, lw, ratios)["mean"], k


def score_from_matrix(x, x2, y, log_lik, reff=1.0, permutations=1, scale=False):
    """Pointwise scores and the Pareto k of ``psislw(-log_lik)`` from ``(n_obs, n_draws)`` matrices (``x2`` may be ``x``) and
    ``y`` of length n_obs; everything of one kind (NumPy, or torch CUDA tensors)."""
    eng = get_engine()
    n, s = x.shape
    torchy = _is_torch_tensor(x)
    exx = None
    for _ in range(permutations):
        shuffle = np.random.permutation(s)  # loo_score.py:305
        if torchy:
            import torch

            idx = torch.as_tensor(shuffle, device=x.device)
            other, ll2 = x2.index_select(1, idx), log_lik.index_select(1, idx)
        else:
            other, ll2 = x2[:, shuffle], log_lik[:, shuffle]
        joint = -log_lik - ll2                                   # 310
        term, _ = _weighted_mean(eng, abs(x - other), joint, reff)  # 311-320
        exx = term if exx is None else exx + term
    exx = exx / permutations                                     # 225
    ycol = y.reshape(n, 1) if torchy else np.asarray(y, dtype=x.dtype).reshape(n, 1)
    exy, k = _weighted_mean(eng, abs(x - ycol), -log_lik, reff)  # 227-237
    if torchy:
        exx, exy, k = exx.cpu().numpy(), exy.cpu().numpy(), k.cpu().numpy()
    return _crps(np.asarray(exx, dtype=np.float64), np.asarray(exy, dtype=np.float64), scale), k


def _validate(x, x2, y, obs_shape):
    """loo_score.py:349-414 on the stacked arrays."""
    if x.shape != x2.shape:
        raise ValueError("x and x2 must have the same shape")
    if np.isnan(x).any() or np.isnan(x2).any() or np.isnan(y).any():
        warnings.warn("NaN values detected in input data. These may lead to unreliable results.", UserWarning, stacklevel=3)
    if np.isinf(x).any() or np.isinf(x2).any() or np.isinf(y).any():
        warnings.warn("Infinite values detected in input data. These may lead to unreliable results.", UserWarning, stacklevel=3)
    if tuple(np.shape(y)) != tuple(obs_shape):
        raise ValueError(f"y dimensions {tuple(np.shape(y))} are not compatible with x dimensions {tuple(obs_shape)}")


def loo_score(data, x_group="posterior_predictive", x_var=None, x2_group=None, x2_var=None, y_group="observed_data",
              y_var=None, var_name=None, pointwise=None, permutations=1, reff=None, scale=False, **kwargs):
    """loo_score.py:48-274.  ``data``: InferenceData (with ArviZ), or without it a dict of groups
    ``{"posterior_predictive": {name: (chain, draw, *obs)}, "observed_data": {name: (*obs)}, "log_likelihood": {...}}``."""
    if kwargs:
        raise TypeError(f"unexpected arguments for e_loo: {sorted(kwargs)}")
    idata = to_inference_data(data)
    log_likelihood = get_log_likelihood(idata, var_name=var_name)
    pointwise = rcParams["stats.ic_pointwise"] if pointwise is None else pointwise
    x_da, x_name = group_variable(idata, x_group, x_var, "x_var")
    x2_da, _ = group_variable(idata, x2_group or x_group, x2_var or x_name, "x2_var")
    y_da, _ = group_variable(idata, y_group, y_var, "y_var")
    x, obs_shape, obs_dims, coords = stack_samples(x_da)
    x2, _, _, _ = stack_samples(x2_da)
    ll, ll_shape, _, _ = stack_samples(log_likelihood)
    y = np.asarray(getattr(y_da, "values", y_da), dtype=np.float64)
    _validate(x, x2, y, obs_shape)
    if tuple(ll_shape) != tuple(obs_shape) or ll.shape != x.shape:
        raise ValueError(f"log_lik dimensions {tuple(ll_shape)} are not compatible with x dimensions {tuple(obs_shape)}")
    n_samples = x.shape[-1]
    if reff is None:  # loo_score.py:202-217
        from .loo import _relative_efficiency

        reff = _relative_efficiency(idata, n_samples)
    dt = np.result_type(x.dtype, ll.dtype)
    xs, x2s, lls = (np.ascontiguousarray(a, dtype=dt) for a in (x, x2, ll))
    score_pw, k = score_from_matrix(xs, xs if x2_da is x_da else x2s, y.reshape(-1), lls, reff, permutations, scale)
    value = float(score_pw.mean())                                   # 241-242
    se = float(score_pw.std() / np.sqrt(score_pw.size))
    estimates = np.array([value, se])
    estimates.dtype = np.dtype([("Estimate", float), ("SE", float)])  # 244-246
    result = LooScoreResult(estimates=estimates, pointwise=score_pw.reshape(obs_shape))
    if pointwise:                                                     # 253-272
        good_k = min(1 - 1 / np.log10(n_samples), 0.7)
        result.pareto_k = wrap_obs(k, obs_shape, obs_dims, coords, "pareto_shape")
        result.good_k = good_k
        if np.any(k > good_k):
            warnings.warn(
                f"Estimated shape parameter of Pareto distribution is greater than {good_k:.2f} for {np.sum(k > good_k)} "
                "observations. This indicates that importance sampling may be unreliable because the marginal posterior and "
                "LOO posterior are very different.", UserWarning, stacklevel=2)
            result.warning = True
        else:
            result.warning = False
    return result
