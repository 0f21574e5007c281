"""``loo_i()`` -- leave-one-out for a single observation, with the reference's signature, warnings,
exceptions and result layout (pyloo/loo_i.py:16-294).  The importance weights of the one row come from the
HIP engine (``compute_importance_weights`` -> ``pla_importance_weights``); the standard error of a single
pointwise value needs the weights themselves (loo_i.py:221-230), which is why this front uses the
weights-returning entry point rather than the fused LOO pass."""

import numpy as np

from ._capi import AGG_COUNT, AGG_MIN_DIAG, AGG_N_HIGH
from .base import ISMethod, compute_importance_weights
from .elpd import ELPDData
from .loo import _checked_method, _diagnostic_warning, _relative_efficiency, _replace_nan, _scale_value
from .rcparams import rcParams
from .utils import get_log_likelihood, stack_samples, to_inference_data, wrap_obs

__all__ = ["loo_i"]


def _lse(v, b_inv=None):
    """utils.py:305-359 for one row (host arithmetic on S values; the heavy part ran on the GPU)."""
    m = np.max(v)
    s = np.log(np.sum(np.exp(v - m))) + m
    return s - np.log(b_inv) if b_inv is not None else s


def loo_i(i, data, pointwise=None, var_name=None, reff=None, scale=None, method="psis"):
    idata = to_inference_data(data)
    log_likelihood = get_log_likelihood(idata, var_name=var_name)
    pointwise = rcParams["stats.ic_pointwise"] if pointwise is None else pointwise
    matrix, obs_shape, obs_dims, coords = stack_samples(log_likelihood)  # loo_i.py:97
    n_samples = matrix.shape[-1]
    n_data_points = 1  # loo_i.py:100
    if isinstance(i, (list, tuple, np.ndarray)):
        raise ValueError("loo_i only accepts a single integer index")
    try:
        i = int(i)
    except (TypeError, ValueError):
        raise TypeError("Index i must be an integer")
    total_obs = int(np.prod(obs_shape))
    if i >= total_obs or i < 0:
        raise IndexError(f"Index {i} is out of bounds for log likelihood array with {total_obs} observations")
    ll_i = np.array(matrix.reshape(total_obs, n_samples)[i], dtype=np.float64)  # one row, copied

    scale, scale_value = _scale_value(scale)
    if reff is None:
        reff = _relative_efficiency(idata, n_samples)
    ll_i = _replace_nan(ll_i)          # loo_i.py:155-164
    method = _checked_method(method)   # loo_i.py:166-181

    lw, diagnostic = compute_importance_weights(-ll_i[None, :], method=method, reff=reff)  # loo_i.py:183-185
    log_weights = np.asarray(lw, dtype=np.float64) + ll_i[None, :]                         # loo_i.py:186
    diagnostic = np.asarray(diagnostic, dtype=np.float64)

    good_k = min(1 - 1 / np.log10(n_samples), 0.7)
    agg = np.zeros(AGG_COUNT)  # the diagnostic summary the shared warning helper reads (loo_i.py:191-212)
    agg[AGG_N_HIGH] = np.sum(diagnostic > good_k)
    agg[AGG_MIN_DIAG] = np.min(diagnostic)
    warn_mg = _diagnostic_warning(method, agg, good_k, n_samples)

    with np.errstate(all="ignore"):
        loo_lppd_i = scale_value * np.array([_lse(log_weights[0])])  # loo_i.py:214-217
        loo_lppd = loo_lppd_i.sum()
        weights = np.exp(log_weights - np.max(log_weights, axis=-1, keepdims=True))  # loo_i.py:219-222
        weights /= np.sum(weights, axis=-1, keepdims=True)
        w2 = weights**2
        lik = np.exp(ll_i[None, :])
        e_epd = np.exp(loo_lppd)
        var_epd = np.sum(w2 * (lik - e_epd) ** 2) / reff                              # loo_i.py:227
        loo_lppd_se = np.sqrt(np.log1p(var_epd / e_epd**2))                           # loo_i.py:228
        lppd = _lse(ll_i, b_inv=n_samples)                                            # loo_i.py:230-238
    p_loo = lppd - loo_lppd / scale_value

    if not pointwise:
        data_, index = [loo_lppd, loo_lppd_se, p_loo, n_samples, n_data_points, warn_mg, scale], \
            ["elpd_loo", "se", "p_loo", "n_samples", "n_data_points", "warning", "scale"]
        if method == ISMethod.PSIS:
            data_.append(good_k)
            index.append("good_k")
        out = ELPDData(data=data_, index=index)
        out.method = method.value
        return out
    loo_da = wrap_obs(loo_lppd_i, (1,), ("_dummy",), {}, "loo_i")
    data_ = [loo_lppd, loo_lppd_se, p_loo, n_samples, n_data_points, warn_mg, loo_da, scale]
    index = ["elpd_loo", "se", "p_loo", "n_samples", "n_data_points", "warning", "loo_i", "scale"]
    if method == ISMethod.PSIS:
        data_ += [diagnostic, good_k]
        index += ["pareto_k", "good_k"]
    else:
        data_.append(diagnostic)
        index.append("ess")
    out = ELPDData(data=data_, index=index)
    out.method = method.value
    return out
