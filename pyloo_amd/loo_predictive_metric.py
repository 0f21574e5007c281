"""``loo_predictive_metric()`` -- leave-one-out MAE / MSE / RMSE / accuracy / balanced accuracy with the reference's
signature and result (pyloo/loo_predictive_metric.py:22-231), on the HIP engine.

The reference calls ``psislw(-log_lik, reff=r_eff)`` (208) and ``e_loo(..., type="mean")`` (210-218) -- one ``_psislw`` and
one weighted mean per observation in Python loops -- and then reduces the n LOO predictions against ``y`` on the host
(234-356).  Here the first two are ``pla_importance_weights`` and ``pla_e_loo`` on the stacked ``(n_obs, n_draws)``
matrices; the five closed forms stay NumPy.  ``predictive_metric_from_matrix`` takes the matrices directly (NumPy, or
torch CUDA tensors, which never leave the device until the n predictions come back)."""

import numpy as np

from .base import tail_count_for
from .engine import _is_torch_tensor, get_engine
from .utils import group_variable, stack_samples, to_inference_data

__all__ = ["loo_predictive_metric", "predictive_metric_from_matrix"]

METRICS = ("mae", "mse", "rmse", "acc", "balanced_acc")


def _same_length(y, yhat):
    if len(y) != len(yhat):
        raise ValueError("y and yhat must have the same length")  # loo_predictive_metric.py:359-363
    return len(y)


def _binary_inputs(y, yhat):
    """loo_predictive_metric.py:366-372."""
    if not np.all((y <= 1) & (y >= 0)):
        raise ValueError("y must contain values between 0 and 1")
    if not np.all((yhat <= 1) & (yhat >= 0)):
        raise ValueError("yhat must contain values between 0 and 1")


def _mean_and_se(e, n):
    return {"estimate": np.mean(e), "se": np.std(e, ddof=1) / np.sqrt(n)}


def _mae(y, yhat):
    """loo_predictive_metric.py:234-252."""
    n = _same_length(y, yhat)
    return _mean_and_se(np.abs(y - yhat), n)


def _mse(y, yhat):
    """loo_predictive_metric.py:255-273."""
    n = _same_length(y, yhat)
    return _mean_and_se((y - yhat) ** 2, n)


def _rmse(y, yhat):
    """loo_predictive_metric.py:276-298: delta method on the MSE."""
    mse = _mse(y, yhat)
    return {"estimate": np.sqrt(mse["estimate"]), "se": np.sqrt(mse["se"] ** 2 / mse["estimate"] / 4)}


def _accuracy(y, yhat):
    """loo_predictive_metric.py:301-326."""
    n = _same_length(y, yhat)
    _binary_inputs(y, yhat)
    est = np.mean(((yhat > 0.5).astype(int) == y).astype(int))
    return {"estimate": est, "se": np.sqrt(est * (1 - est) / n)}


def _balanced_accuracy(y, yhat):
    """loo_predictive_metric.py:329-356."""
    n = _same_length(y, yhat)
    _binary_inputs(y, yhat)
    hit = (yhat > 0.5).astype(int) == y
    neg = y == 0
    tn, tp = np.mean(hit[neg]), np.mean(hit[~neg])
    return {"estimate": (tp + tn) / 2, "se": np.sqrt((tp * (1 - tp) + tn * (1 - tn)) / 4 / n)}


_REDUCERS = {"mae": _mae, "mse": _mse, "rmse": _rmse, "acc": _accuracy, "balanced_acc": _balanced_accuracy}


def _check_metric(metric):
    if metric not in METRICS:
        raise ValueError(f"Invalid metric: {metric}. Must be one of: 'mae', 'mse', 'rmse', 'acc', 'balanced_acc'")


def loo_predictions(x, log_lik, r_eff=1.0):
    """PSIS-LOO predictive means of the draws ``x`` -- loo_predictive_metric.py:208-220 -- for ``(n_obs, n_draws)`` matrices
    (NumPy or torch CUDA): smoothed weights of ``-log_lik``, then their weighted mean of ``x``.  Returns (mean, pareto_k of the
    mean)."""
    if tuple(x.shape) != tuple(log_lik.shape):
        raise ValueError(f"predictions {tuple(x.shape)} and log-likelihood {tuple(log_lik.shape)} must have the same shape")
    eng = get_engine()
    ratios = -log_lik
    lw, _ = eng.importance_weights(ratios, tail_count_for(ratios.shape[-1], r_eff), "psis")
    res = eng.e_loo(x if x.dtype == lw.dtype else (x.to(lw.dtype) if _is_torch_tensor(x) else x.astype(lw.dtype)), lw, ratios)
    return res["mean"], res["k_mean"]


def predictive_metric_from_matrix(x, log_lik, y, metric="mae", r_eff=1.0):
    """The metric from ``(n_obs, n_draws)`` matrices of predictive draws and pointwise log-likelihoods."""
    _check_metric(metric)
    y = np.asarray(y).flatten()
    if len(y) != x.shape[0]:
        raise ValueError(f"Length of y ({len(y)}) must match the number of observations in x ({x.shape[0]})")
    pred, _ = loo_predictions(x, log_lik, r_eff)
    if _is_torch_tensor(pred):
        pred = pred.cpu().numpy()
    return _REDUCERS[metric](y, np.asarray(pred, dtype=np.float64))


def loo_predictive_metric(data, y, var_name=None, group="posterior_predictive", log_lik_group="log_likelihood",
                          log_lik_var_name=None, metric="mae", r_eff=1.0, **kwargs):
    """loo_predictive_metric.py:22-231.  ``data``: InferenceData (with ArviZ), or without it a dict of groups
    ``{"posterior_predictive": {name: (chain, draw, *obs)}, "log_likelihood": {name: (chain, draw, *obs)}}``.
    Returns ``{"estimate": ..., "se": ...}``."""
    if kwargs:
        unknown = set(kwargs) - {"type"}  # (what the reference forwards to e_loo; it always asks for the mean)
        if unknown:
            raise TypeError(f"unexpected arguments for e_loo: {sorted(unknown)}")
    y = np.asarray(y).flatten()
    idata = to_inference_data(data)
    if not hasattr(idata, group):
        raise ValueError(f"InferenceData object does not have a {group} group")
    if not hasattr(idata, log_lik_group):
        raise ValueError(f"InferenceData object does not have a {log_lik_group} group")
    log_lik, _ = group_variable(idata, log_lik_group, log_lik_var_name, "log_lik_var_name")
    x, _ = group_variable(idata, group, var_name)
    xm, obs_shape, _, _ = stack_samples(x)
    lm, _, _, _ = stack_samples(log_lik)
    n_obs = obs_shape[0] if obs_shape else 1  # loo_predictive_metric.py:193-194: the first observation dimension
    if len(y) != n_obs:
        raise ValueError(f"Length of y ({len(y)}) must match the number of observations in x ({n_obs})")
    _check_metric(metric)
    pred, _ = loo_predictions(np.ascontiguousarray(xm), np.ascontiguousarray(lm), r_eff)
    return _REDUCERS[metric](y, np.asarray(pred, dtype=np.float64).reshape(obs_shape))
