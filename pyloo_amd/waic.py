"""``waic()`` -- widely applicable information criterion with the reference's signature and result
object (pyloo/waic.py:16-207), executed by the HIP engine.

Host Python: argument handling, the NaN / inf / variance warnings and ``ELPDData`` packing
(waic.py:90-135,147-207).  Engine (``pla_waic``, one read of the matrix): the replacements of
waic.py:112-135, ``lppd_i`` (137-143), the variance over draws (145), ``waic_i`` and the sums (158-161).
"""

import warnings

import numpy as np

from ._capi import AGG_M2_LOO, AGG_N_HIGH, AGG_N_SLOW, AGG_SUM_LOO, AGG_SUM_LPPD
from .elpd import ELPDData
from .engine import _is_torch_tensor, get_engine
from .loo import _scale_value
from .rcparams import rcParams
from .sharded import all_reduce_aggregates
from .utils import get_log_likelihood, stack_samples, to_inference_data, wrap_obs

__all__ = ["waic", "waic_from_matrix"]


def _warn_replaced(matrix):
    """waic.py:109-135: the reference warns separately for NaN and for infinite entries (the engine
    applies the replacements itself while it reads the matrix)."""
    if _is_torch_tensor(matrix):
        import torch

        has_nan, has_inf = bool(torch.isnan(matrix).any()), bool(torch.isinf(matrix).any())
    else:
        has_nan, has_inf = bool(np.isnan(matrix).any()), bool(np.isinf(matrix).any())
    if has_nan:
        warnings.warn(
            "NaN values detected in log-likelihood. These will be ignored in the WAIC calculation.",
            UserWarning,
            stacklevel=3,
        )
    if has_inf:
        warnings.warn(
            "Infinite values detected in log-likelihood. These will be ignored in the WAIC calculation.",
            UserWarning,
            stacklevel=3,
        )


def _finish(res, agg, n_samples, n_data_points, scale, pointwise, wrap=None):
    warn_mg = bool(agg[AGG_N_HIGH] > 0)  # waic.py:147
    if warn_mg:
        warnings.warn(
            "For one or more samples the posterior variance of the log predictive "
            "densities exceeds 0.4. This could be indication of WAIC starting to fail.",
            UserWarning,
            stacklevel=3,
        )
    waic_sum = float(agg[AGG_SUM_LOO])       # waic.py:160
    waic_se = float(agg[AGG_M2_LOO]) ** 0.5  # waic.py:159: (n * var)^0.5 with var = M2 / n
    p_waic = float(agg[AGG_SUM_LPPD])        # waic.py:161
    if not pointwise:
        return ELPDData(
            data=[waic_sum, waic_se, p_waic, n_samples, n_data_points, warn_mg, scale],
            index=["elpd_waic", "se", "p_waic", "n_samples", "n_data_points", "warning", "scale"],
        )
    waic_i = res["waic_i"]
    wi = waic_i.detach().cpu().numpy() if hasattr(waic_i, "detach") else np.asarray(waic_i)
    if wi.size and np.allclose(wi, wi.flat[0]):  # waic.py:178-184
        warnings.warn(
            "The point-wise WAIC is the same with the sum WAIC, please double check "
            "the Observed RV in your model to make sure it returns element-wise logp.",
            UserWarning,
            stacklevel=3,
        )
    return ELPDData(
        data=[waic_sum, waic_se, p_waic, n_samples, n_data_points, warn_mg, wrap(wi) if wrap else waic_i, scale],
        index=["elpd_waic", "se", "p_waic", "n_samples", "n_data_points", "warning", "waic_i", "scale"],
    )


def waic_from_matrix(log_likelihood, scale=None, pointwise=False, distributed=False, group=None):
    """WAIC from an ``(n_obs, n_draws)`` matrix: NumPy array (host) or torch CUDA tensor (device-resident,
    read once).  ``distributed=True``: every rank passes its own block of observations; the sums are merged
    with the same single all-reduce as ``loo_from_matrix``."""
    scale, scale_value = _scale_value(scale)
    n_local, n_samples = log_likelihood.shape
    dev = log_likelihood.device.index if _is_torch_tensor(log_likelihood) else None
    res = get_engine(dev).waic(log_likelihood, scale_value)
    if distributed:
        agg = all_reduce_aggregates(res["agg"], group)
    else:
        a = res["agg"]
        agg = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    if agg[AGG_N_SLOW] > 0:
        _warn_replaced(log_likelihood)
    return _finish(res, agg, n_samples, int(agg[0]), scale, pointwise)


def waic(data, pointwise=None, var_name=None, scale=None):
    """Widely applicable information criterion.  Same parameters, warnings, exceptions and ``ELPDData``
    layout as ``pyloo.waic`` (waic.py:16-207)."""
    idata = to_inference_data(data)
    log_likelihood = get_log_likelihood(idata, var_name=var_name)
    pointwise = rcParams["stats.ic_pointwise"] if pointwise is None else pointwise
    matrix, obs_shape, obs_dims, coords = stack_samples(log_likelihood)  # waic.py:93
    n_samples = matrix.shape[-1]
    n_data_points = int(np.prod(obs_shape))
    scale, scale_value = _scale_value(scale)
    res = get_engine(None).waic(matrix, scale_value)
    agg = np.asarray(res["agg"])
    if agg[AGG_N_SLOW] > 0:
        _warn_replaced(matrix)
    return _finish(res, agg, n_samples, n_data_points, scale, pointwise,
                   wrap=lambda w: wrap_obs(w, obs_shape, obs_dims, coords, "waic_i"))
