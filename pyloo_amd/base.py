"""``compute_importance_weights`` -- the importance-sampling dispatcher of the reference
(pyloo/base.py:29-175) on top of the HIP engine: one batched C-ABI call replaces the per-row
Python loop that base.py:160-166 hands to ``wrap_xarray_ufunc``."""

from enum import Enum

import numpy as np

from .engine import _is_torch_tensor, get_engine

try:
    import xarray as xr
except Exception:  # pragma: no cover
    xr = None

__all__ = ["ISMethod", "compute_importance_weights"]


class ISMethod(str, Enum):
    """Supported importance sampling methods (base.py:18-23)."""

    PSIS = "psis"
    SIS = "sis"
    TIS = "tis"


def parse_method(method):
    if isinstance(method, ISMethod):
        return method
    try:
        return ISMethod(method.lower())
    except (ValueError, AttributeError):
        valid = ", ".join(m.value for m in ISMethod)
        raise ValueError(f"Invalid method '{method}'. Must be one of: {valid}") from None


def tail_count_for(n_samples, reff):
    """M with ``cutoff_ind = -M - 1``: same Python expression as base.py:139-141, so invalid
    ``reff`` values raise exactly what they raise in the reference."""
    return int(np.ceil(min(n_samples / 5.0, 3 * (n_samples / reff) ** 0.5)))


def _run(values, method, reff):
    """values: ndarray or CUDA tensor (..., S) -> (lw same shape/dtype, diag lead-shape float64)."""
    torchy = _is_torch_tensor(values)
    shape = tuple(values.shape)
    if len(shape) < 1:
        raise ValueError("log_weights needs a sample dimension")
    n_samples = shape[-1]
    lead = shape[:-1]
    M = tail_count_for(n_samples, reff) if method == ISMethod.PSIS else 0
    if method == ISMethod.PSIS and M + 1 > n_samples:
        # x_sort_ind[cutoff_ind] of psis.py:136 is out of range
        raise IndexError(f"index {-M - 1} is out of bounds for axis 0 with size {n_samples}")
    n_obs = int(np.prod(lead)) if lead else 1
    if torchy:
        flat = values.reshape(n_obs, n_samples)
        lw, diag = get_engine(flat.device.index).importance_weights(flat, M, method.value)
        return lw.reshape(shape), diag.reshape(lead)
    arr = np.asarray(values)
    if arr.dtype not in (np.float32, np.float64):
        arr = arr.astype(np.float64)
    if n_obs == 0:
        return np.empty_like(arr), np.empty(lead)
    lw, diag = get_engine().importance_weights(arr.reshape(n_obs, n_samples), M, method.value)
    return lw.reshape(shape), np.asarray(diag.reshape(lead))


def compute_importance_weights(log_weights=None, method=ISMethod.PSIS, reff=1.0):
    """Smoothed / truncated / normalised log weights and the method's diagnostic.

    Same contract as base.py:29-175: the last dimension (or the ``__sample__`` dimension of a
    DataArray) holds the draws; returns ``(lw, diagnostic)`` with ``lw`` shaped like the input
    and the diagnostic (Pareto k for PSIS, ESS for SIS/TIS) shaped like the leading dims --
    a 0-d array for 1-D input.  The input is never modified.
    """
    if xr is not None and isinstance(log_weights, xr.DataArray):
        if "__sample__" not in log_weights.dims:
            if "chain" in log_weights.dims and "draw" in log_weights.dims:
                log_weights = log_weights.stack(__sample__=("chain", "draw"))
            else:
                raise ValueError("log_weights must have a __sample__ dimension")
    method = parse_method(method)
    if log_weights is None:
        raise ValueError("log_weights must be provided when variational=False")
    if xr is not None and isinstance(log_weights, xr.DataArray):
        da = log_weights.transpose(..., "__sample__")
        lw, diag = _run(da.values, method, reff)
        obs_dims = da.dims[:-1]
        lw_da = xr.DataArray(lw, dims=da.dims, coords=da.coords, name="log_weights").transpose(*log_weights.dims)
        diag_da = xr.DataArray(diag, dims=obs_dims, coords={d: da.coords[d] for d in obs_dims if d in da.coords},
                               name="pareto_shape" if method == ISMethod.PSIS else "ess")
        return lw_da, diag_da
    return _run(log_weights, method, reff)
