"""``e_loo()`` -- PSIS-weighted expectations of posterior(-predictive) draws and their function-specific Pareto k, with
the reference's signature and result object (pyloo/e_loo.py:24-264), executed by the HIP engine.

Host Python: argument handling and error texts (e_loo.py:150-213), the three closed-form diagnostics of the k values
(393-427) and ``ExpectationResult`` packing.  Engine (``pla_e_loo``): per observation the normalised weights, weighted
mean / variance (430-465, 518-531, 557-559) and ``k_hat`` (328-390) -- reproduced as the reference evaluates it, see
``csrc/pla_eloo.h``.

``type="quantile"`` (468-515, 534-554) runs on the device as well (``pla_e_loo_quantiles``: a weighted radix selection per
observation and probability instead of the reference's argsort); the result then carries a trailing ``quantile`` axis.
ArviZ / xarray are optional here: InferenceData / DataArray inputs are taken when those packages are importable, plain
arrays ``(chain, draw, *obs)`` or ``(*obs, n_draws)`` otherwise.
"""

from dataclasses import dataclass
from typing import Any

import numpy as np

from .engine import _is_torch_tensor, get_engine
from .utils import to_inference_data, wrap_obs, xr

__all__ = ["e_loo", "ExpectationResult", "compute_pareto_k", "k_hat", "_pareto_min_ss", "_pareto_khat_threshold",
           "_pareto_convergence_rate"]


@dataclass
class ExpectationResult:
    """e_loo.py:24-53: ``value``, ``pareto_k`` and the three reliability diagnostics (same shape as the observations)."""

    value: Any
    pareto_k: Any
    min_ss: Any = None
    khat_threshold: Any = None
    convergence_rate: Any = None


def _pareto_min_ss(k):
    """e_loo.py:393-398."""
    if k < 1:
        return 10 ** (1 / (1 - max(0, k)))
    return float("inf")


def _pareto_khat_threshold(n_samples):
    """e_loo.py:401-403."""
    return 1 - 1 / np.log10(n_samples)


def _pareto_convergence_rate(k, n_samples):
    """e_loo.py:406-427."""
    if k < 0:
        return 1.0
    if k > 1:
        return 0.0
    if k == 0.5:
        return 1 - 1 / np.log(n_samples)
    if 0 < k < 1:
        n = n_samples
        return max(0, (2 * (k - 1) * n ** (2 * k + 1) + (1 - 2 * k) * n ** (2 * k) + n**2) / ((n - 1) * (n - n ** (2 * k))))
    return 1.0


def _sample_last(a, what):
    """``(*obs, n_draws)`` view of an input: DataArrays are stacked like e_loo.py:198-211, ``(chain, draw, *obs)`` arrays
    cannot be told from ``(*obs, n_draws)`` ones by shape, so plain arrays must already have the draws last."""
    if xr is not None and isinstance(a, xr.DataArray):
        if "__sample__" not in a.dims:
            if "chain" in a.dims and "draw" in a.dims:
                a = a.stack(__sample__=("chain", "draw"))
            else:
                a = a.rename({a.dims[-1]: "__sample__"})
        a = a.transpose(..., "__sample__")
        dims = tuple(d for d in a.dims if d != "__sample__")
        return a.values, dims, {d: a.coords[d] for d in dims if d in a.coords}
    if _is_torch_tensor(a):
        return a, None, {}
    a = np.asarray(a)
    if a.ndim < 1:
        raise ValueError(f"{what} must have a sample dimension")
    return a, None, {}


def _obs_dims(a):
    return tuple(d for d in a.dims if d not in ("__sample__", "chain", "draw"))


def _like(a, ref, what):
    """``a`` with its observation dims in ``ref``'s order when both are DataArrays (same set of names, or ValueError)."""
    if a is None or xr is None or not (isinstance(a, xr.DataArray) and isinstance(ref, xr.DataArray)):
        return a
    want, have = _obs_dims(ref), _obs_dims(a)
    if set(want) != set(have):
        raise ValueError(f"{what} has dims {tuple(a.dims)}, the data {tuple(ref.dims)}: the observation dims must have the same names")
    if want == have:
        return a
    rest = tuple(d for d in a.dims if d not in have)
    return a.transpose(*want, *rest)


def _rows(a):
    return a.reshape(-1, a.shape[-1])


def compute_pareto_k(x, log_ratios, tail_len=20):
    """e_loo.py:266-325: k of h = ``x`` (or of the ratios alone when ``x`` is None) for every observation; arrays have the
    draws on the last axis.  1-D input gives a float, like the reference's ndarray branch."""
    if tail_len < 5:
        raise ValueError("tail_len must be at least 5")
    lr, dims, coords = _sample_last(log_ratios, "log_ratios")
    if x is not None:
        xv, _, _ = _sample_last(x, "x")
        if tuple(xv.shape) != tuple(lr.shape):
            raise ValueError("x and log_ratios must have the same shape")
    else:
        xv = None
    eng = get_engine()
    lr2 = _rows(lr)
    res = eng.e_loo(_rows(xv) if xv is not None else lr2, lr2, None, tail_len)
    k = res["k_mean"] if xv is not None else res["k_none"]
    if _is_torch_tensor(k):
        return k.reshape(tuple(lr.shape[:-1]))
    if lr.ndim == 1:
        return float(k[0])
    return wrap_obs(k, lr.shape[:-1], dims or (), coords, "pareto_k") if dims else k.reshape(lr.shape[:-1])


def k_hat(x_vals, log_ratios_vals, tail_len=20):
    """e_loo.py:328-390 for one observation."""
    return compute_pareto_k(x_vals, np.asarray(log_ratios_vals), tail_len)


def e_loo(data, var_name=None, group="posterior_predictive", weights=None, log_weights=None, log_ratios=None, type="mean",
          probs=None):
    """e_loo.py:56-264.  ``data``: InferenceData / DataArray (with ArviZ / xarray), or an array of draws with the sample
    axis LAST, shape ``(*obs, n_draws)``; the weights in the same layout."""
    if type not in ["mean", "variance", "sd", "quantile"]:
        raise ValueError("type must be 'mean', 'variance', 'sd' or 'quantile'")
    if type == "quantile":
        if probs is None:
            raise ValueError("probs must be provided for quantile calculation")
        probs_array = np.array([probs]) if np.isscalar(probs) else np.asarray(probs)
        if not np.all((probs_array > 0) & (probs_array < 1)):
            raise ValueError("probs must be between 0 and 1")
    if weights is None and log_weights is None:
        raise ValueError("Either weights or log_weights must be provided")

    if (xr is not None and isinstance(data, xr.DataArray)) or isinstance(data, np.ndarray) or _is_torch_tensor(data):
        x_data = data
    else:
        idata = to_inference_data(data)
        if not hasattr(idata, group):
            raise ValueError(f"InferenceData object does not have a {group} group")
        data_group = getattr(idata, group)
        names = list(data_group.data_vars) if hasattr(data_group, "data_vars") else list(data_group.keys())
        if var_name is None:
            if len(names) == 1:
                var_name = names[0]
            else:
                raise ValueError(f"Multiple variables found in {group} group. Please specify var_name from: {names}")
        elif var_name not in names:
            raise ValueError(f"Variable '{var_name}' not found in {group} group. Available variables: {names}")
        x_data = data_group[var_name]

    if weights is not None:  # e_loo.py:202-203
        if _is_torch_tensor(weights):
            log_weights = weights.log()
        elif xr is not None and isinstance(weights, xr.DataArray):
            log_weights = np.log(weights)
        else:
            with np.errstate(divide="ignore"):
                log_weights = np.log(np.asarray(weights))
    xv, dims, coords = _sample_last(x_data, "data")
    # DataArrays pair up by dimension NAME (the reference aligns and broadcasts by name, e_loo.py:205-212): the weights and
    # ratios are brought to the data's observation dims before their values are taken; other names are an error
    log_weights = _like(log_weights, x_data, "log_weights")
    log_ratios = _like(log_ratios, x_data, "log_ratios")
    lw, _, _ = _sample_last(log_weights, "log_weights")
    lr = None if log_ratios is None else _sample_last(log_ratios, "log_ratios")[0]
    if tuple(lw.shape) != tuple(xv.shape) or (lr is not None and tuple(lr.shape) != tuple(xv.shape)):
        raise ValueError(f"data {tuple(xv.shape)} and the weights {tuple(lw.shape)} must have the same shape (draws last)")
    n_samples = xv.shape[-1]
    obs_shape = tuple(xv.shape[:-1])
    res = get_engine().e_loo(_rows(xv), _rows(lw), None if lr is None else _rows(lr))
    if type == "quantile":
        q = get_engine().e_loo_quantiles(_rows(xv), _rows(lw), probs_array)               # e_loo.py:468-515
        value, k = q, res["k_none"]                                                       # 229-230: h = None
    elif type == "mean":
        value, k = res["mean"], res["k_mean"]
    elif type == "variance":
        value, k = res["var"], res["k_var"]
    else:
        value, k = res["var"] ** 0.5, res["k_var"]

    if type == "quantile":  # value: (*obs, quantile) (e_loo.py:509-515)
        qshape = obs_shape + (len(probs_array),)
        if xr is not None and dims:
            value = xr.DataArray(np.asarray(value).reshape(qshape), dims=tuple(dims) + ("quantile",),
                                 coords={**coords, "quantile": probs_array})
        else:
            value = value.reshape(qshape)
    if _is_torch_tensor(k):
        kh = k.cpu().numpy()
        shape = lambda a: a.reshape(obs_shape)  # noqa: E731
    else:
        kh = k
        shape = (lambda a: wrap_obs(a, obs_shape, dims, coords, None)) if dims else (lambda a: np.asarray(a).reshape(obs_shape))
    min_ss = np.array([_pareto_min_ss(v) for v in kh])                                   # e_loo.py:243
    rate = np.array([_pareto_convergence_rate(v, n_samples) for v in kh])               # 246-250
    thr = np.full(kh.shape, _pareto_khat_threshold(n_samples))                           # 244
    plain = (lambda a: wrap_obs(a, obs_shape, dims, coords, None)) if dims else (lambda a: np.asarray(a).reshape(obs_shape))
    return ExpectationResult(value=value if type == "quantile" else shape(value), pareto_k=shape(k), min_ss=plain(min_ss),
                             khat_threshold=plain(thr), convergence_rate=plain(rate))
