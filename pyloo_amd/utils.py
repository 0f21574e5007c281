"""Host-side data access for the hot path: getting an ``(n_obs, n_draws)`` matrix out of
whatever the user passed.  Mirrors pyloo/utils.py:21-79 (``to_inference_data``) and
utils.py:257-302 (``get_log_likelihood``) where ArviZ/xarray are installed, and accepts plain
arrays where they are not (they are optional here; the reference requires them)."""

import warnings

import numpy as np

try:  # optional third-party containers
    import xarray as xr
except Exception:  # pragma: no cover - absent in the build container
    xr = None
try:
    import arviz as az
except Exception:  # pragma: no cover
    az = None


class LogLikelihoodArray:
    """Minimal stand-in for a ``DataArray`` with dims ``(chain, draw, *obs)`` when xarray is absent."""

    def __init__(self, values, name="obs"):
        values = np.asarray(values)
        if values.ndim < 2:
            raise TypeError("log likelihood must have at least (chain, draw) dimensions")
        self.values = values
        self.name = name
        self.dims = ("chain", "draw") + tuple(f"{name}_dim_{i}" for i in range(values.ndim - 2))


class SimpleInferenceData:
    """Duck-typed InferenceData used only when ArviZ is not installed: ``.log_likelihood`` (and any other sample group, e.g.
    ``posterior_predictive``) is a mapping var_name -> LogLikelihoodArray ``(chain, draw, *obs)``; ``.posterior`` and
    ``.observed_data`` are mappings var_name -> ndarray."""

    def __init__(self, log_likelihood=None, posterior=None, observed_data=None, **sample_groups):
        if log_likelihood is not None:
            self.log_likelihood = {k: LogLikelihoodArray(v, k) for k, v in log_likelihood.items()}
        if posterior is not None:
            self.posterior = {k: np.asarray(v) for k, v in posterior.items()}
        if observed_data is not None:
            self.observed_data = {k: np.asarray(v) for k, v in observed_data.items()}
        for group, content in sample_groups.items():
            if content is not None:
                setattr(self, group, {k: LogLikelihoodArray(v, k) for k, v in content.items()})


def group_variable(idata, group, var_name, arg="var_name"):
    """One variable of one group, with the reference's error texts (e_loo.py:170-196, loo_score.py:456-515,
    loo_predictive_metric.py:156-178)."""
    if not hasattr(idata, group):
        raise ValueError(f"InferenceData object does not have a {group} group")
    content = getattr(idata, group)
    names = list(content.data_vars) if hasattr(content, "data_vars") else list(content.keys())
    if var_name is None:
        if len(names) == 1:
            return content[names[0]], names[0]
        raise ValueError(f"Multiple variables found in {group} group. Please specify {arg} from: {names}")
    if var_name not in names:
        raise ValueError(f"Variable '{var_name}' not found in {group} group. Available variables: {names}")
    return content[var_name], var_name


def to_inference_data(obj):
    """utils.py:21-79: anything ArviZ can convert; without ArviZ: dicts / ndarrays / duck types."""
    if az is not None:
        if isinstance(obj, az.InferenceData):
            return obj
        if isinstance(obj, dict) and ("log_likelihood" in obj or "posterior" in obj):
            return az.from_dict(**obj)
        try:
            return az.convert_to_inference_data(obj)
        except Exception as err:  # same contract as the reference: a ValueError with the allowed types
            raise ValueError(f"Can only convert objects ArviZ understands to InferenceData, not {type(obj).__name__}") from err
    if isinstance(obj, SimpleInferenceData) or hasattr(obj, "log_likelihood") or hasattr(obj, "posterior"):
        return obj
    if isinstance(obj, dict):
        return SimpleInferenceData(**obj)
    arr = np.asarray(obj)
    if arr.dtype.kind in "fiu" and arr.ndim >= 3:
        return SimpleInferenceData(log_likelihood={"obs": arr})
    raise ValueError(
        "Without ArviZ installed, pass a dict(log_likelihood={name: array(chain, draw, *obs)}, posterior=...) "
        "or an ndarray of shape (chain, draw, *obs)"
    )


def get_log_likelihood(idata, var_name=None, single_var=True):
    """utils.py:257-302."""
    if (
        not hasattr(idata, "log_likelihood")
        and hasattr(idata, "sample_stats")
        and hasattr(idata.sample_stats, "log_likelihood")
    ):
        warnings.warn("Storing the log_likelihood in sample_stats groups has been deprecated", DeprecationWarning, stacklevel=2)
        return idata.sample_stats.log_likelihood
    if not hasattr(idata, "log_likelihood"):
        raise TypeError("log likelihood not found in inference data object")
    group = idata.log_likelihood
    names = list(group.data_vars) if hasattr(group, "data_vars") else list(group.keys())
    if var_name is None:
        if len(names) > 1:
            if single_var:
                raise TypeError(f"Found several log likelihood arrays {names}, var_name cannot be None")
            return group[names]
        return group[names[0]]
    try:
        return group[var_name]
    except KeyError as err:
        raise TypeError(f"No log likelihood data named {var_name} found") from err


def stack_samples(log_likelihood):
    """``.stack(__sample__=("chain", "draw"))`` of loo.py:189 as plain arrays.

    Returns ``(matrix, obs_shape, obs_dims, obs_coords)`` where ``matrix`` is ``(n_obs, n_draws)``
    with draws contiguous (chain-major, like xarray's stack) in the input's float dtype.
    """
    if xr is not None and isinstance(log_likelihood, xr.DataArray):
        da = log_likelihood
        if "__sample__" not in da.dims:
            da = da.stack(__sample__=("chain", "draw"))
        da = da.transpose(..., "__sample__")
        vals = da.values
        obs_dims = tuple(d for d in da.dims if d != "__sample__")
        coords = {d: da.coords[d] for d in obs_dims if d in da.coords}
    else:
        vals = np.asarray(getattr(log_likelihood, "values", log_likelihood))
        if vals.ndim < 2:
            raise TypeError("log likelihood must have (chain, draw, *obs) dimensions")
        vals = np.moveaxis(vals.reshape((vals.shape[0] * vals.shape[1],) + vals.shape[2:]), 0, -1)
        obs_dims = tuple(getattr(log_likelihood, "dims", ("chain", "draw") + tuple(f"dim_{i}" for i in range(vals.ndim - 1)))[2:])
        coords = {}
    if vals.dtype not in (np.float32, np.float64):
        vals = vals.astype(np.float64)
    obs_shape = vals.shape[:-1]
    # (chain, draw, *obs) in memory gives an (n_obs, n_draws) VIEW with the observations fastest: the engine takes that
    # layout as it is (one pitched copy to the device, transposed there) -- no transposing copy on the host
    matrix = vals.reshape(-1, vals.shape[-1])
    n, s = matrix.shape
    fast = matrix.strides[1] == matrix.itemsize or (n > 1 and s > 1 and matrix.strides[0] == matrix.itemsize
                                                    and matrix.strides[1] >= n * matrix.itemsize)
    if not fast:
        matrix = np.ascontiguousarray(matrix)
    return matrix, obs_shape, obs_dims, coords


def wrap_obs(values, obs_shape, obs_dims, coords, name):
    """Pointwise vectors go back to the observation dims; a DataArray when xarray is there."""
    values = np.asarray(values).reshape(obs_shape)
    if xr is not None and len(obs_dims) == len(obs_shape):
        return xr.DataArray(values, dims=obs_dims, coords=coords, name=name)
    return values
