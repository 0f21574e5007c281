"""``loo()`` -- PSIS leave-one-out cross-validation with the reference's signature and result
object (pyloo/loo.py:20-513), executed by the HIP engine.

What runs where: argument handling, warnings and ``ELPDData`` packing are host Python (they
follow loo.py:179-249,291-304,344-412); everything between ``compute_importance_weights`` and
the final sums (loo.py:286-342: three Python loops over observations in the reference) is one
``pla_psis_loo`` call.  ``mixture=True`` and ``moment_match=True`` are outside this project's
scope and raise ``NotImplementedError``.
"""

import warnings

import numpy as np

from ._capi import AGG_M2_LOO, AGG_N_HIGH, AGG_MIN_DIAG, AGG_SUM_LOO, AGG_SUM_LPPD
from .base import ISMethod, parse_method, tail_count_for
from .elpd import ELPDData
from .engine import _is_torch_tensor, get_engine
from .rcparams import rcParams
from .sharded import all_reduce_aggregates
from .utils import get_log_likelihood, stack_samples, to_inference_data, wrap_obs

__all__ = ["loo", "loo_from_matrix"]

_SCALE_VALUES = {"deviance": -2, "log": 1, "negative_log": -1}  # loo.py:195-200


def _scale_value(scale):
    scale = rcParams["stats.ic_scale"] if scale is None else scale.lower()
    if scale not in _SCALE_VALUES:
        raise TypeError('Valid scale values are "deviance", "log", "negative_log"')
    return scale, _SCALE_VALUES[scale]


def _relative_efficiency(idata, n_samples):
    """loo.py:204-216: 1.0 for one chain, else mean ArviZ ESS of the posterior / n_samples."""
    if not hasattr(idata, "posterior"):
        raise TypeError("Must be able to extract a posterior group from data.")
    posterior = idata.posterior
    if hasattr(posterior, "chain"):
        n_chains = len(posterior.chain)
    else:
        n_chains = next(iter(posterior.values())).shape[0]
    if n_chains == 1:
        return 1.0
    try:
        from arviz.stats.diagnostics import ess
    except Exception as err:  # pragma: no cover - ArviZ absent
        raise TypeError(
            "reff=None needs ArviZ to estimate the effective sample size of the posterior; "
            "pass reff explicitly (e.g. reff=1.0)."
        ) from err
    ess_p = ess(posterior, method="mean")
    return np.hstack([ess_p[v].values.flatten() for v in ess_p.data_vars]).mean() / n_samples


def _engine_pass(matrix, method, reff, scale_value, good_k, distributed=False, group=None):
    """One fused pass + (optionally) the cross-rank merge.  Returns (pointwise dict, agg ndarray)."""
    n_samples = matrix.shape[-1]
    M = tail_count_for(n_samples, reff) if method == ISMethod.PSIS else 0
    if method == ISMethod.PSIS and M + 1 > n_samples:
        raise IndexError(f"index {-M - 1} is out of bounds for axis 0 with size {n_samples}")
    dev = matrix.device.index if _is_torch_tensor(matrix) else None
    res = get_engine(dev).psis_loo(matrix, M, method.value, scale_value, good_k)
    if distributed:
        agg = all_reduce_aggregates(res["agg"], group)
    else:
        a = res["agg"]
        agg = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return res, agg


def _summaries(agg, n_data_points, scale_value):
    """loo.py:326-342 from the reduced moments."""
    elpd = float(agg[AGG_SUM_LOO])
    m2 = float(agg[AGG_M2_LOO])
    se = m2**0.5  # (n * var)^0.5 with var = M2 / n
    lppd = float(agg[AGG_SUM_LPPD])
    return {
        "elpd_loo": elpd,
        "se": se,
        "p_loo": lppd - elpd / scale_value,
        "p_loo_se": (m2 / n_data_points) ** 0.5 if n_data_points else float("nan"),
        "looic": -2 * elpd,
        "looic_se": 2 * se,
    }


def _diagnostic_warning(method, agg, good_k, n_samples):
    """loo.py:291-317.  Returns the ``warning`` flag."""
    if method == ISMethod.PSIS:
        n_high = int(agg[AGG_N_HIGH])
        if n_high > 0:
            warnings.warn(
                "Estimated shape parameter of Pareto distribution is greater than"
                f" {good_k:.2f} for {n_high} observations. This indicates that"
                " importance sampling may be unreliable because the marginal"
                " posterior and LOO posterior are very different.",
                UserWarning,
                stacklevel=3,
            )
            return True
        return False
    min_ess = float(agg[AGG_MIN_DIAG])
    if min_ess < n_samples * 0.1:
        warnings.warn(
            f"Low effective sample size detected (minimum ESS: {min_ess:.1f})."
            " This indicates that the importance sampling approximation may be"
            " unreliable. Consider using PSIS which is more robust to such"
            " cases.",
            UserWarning,
            stacklevel=3,
        )
        return True
    return False


def _replace_nan(values):
    """loo.py:218-227 / loo_i.py:155-164: NaN log-likelihoods count as -1e10, with a warning."""
    nan_mask = np.isnan(values)
    if not nan_mask.any():
        return values
    warnings.warn(
        "NaN values detected in log-likelihood. These will be ignored in the LOO calculation.",
        UserWarning,
        stacklevel=3,
    )
    return np.where(nan_mask, values.dtype.type(-1e10), values)


def _checked_method(method):
    """loo.py:229-244 / loo_i.py:166-181: parse the method, recommend PSIS when something else is asked for."""
    method = parse_method(method)
    if method != ISMethod.PSIS:
        warnings.warn(
            f"Using {method.value.upper()} for LOO computation. Note that PSIS is the"
            " recommended method as it is typically more efficient and reliable.",
            UserWarning,
            stacklevel=3,
        )
    return method


def _pack(summ, n_samples, n_data_points, warn, scale, method, good_k, pointwise, loo_i=None, diag=None):
    """Index order of loo.py:516-626 + 360-365 / 400-410."""
    data = [summ["elpd_loo"], summ["se"], summ["p_loo"], summ["p_loo_se"], n_samples, n_data_points, warn]
    index = ["elpd_loo", "se", "p_loo", "p_loo_se", "n_samples", "n_data_points", "warning"]
    if pointwise:
        data.append(loo_i)
        index.append("loo_i")
    data += [scale, summ["looic"], summ["looic_se"]]
    index += ["scale", "looic", "looic_se"]
    if pointwise:
        data.append(diag)
        index.append("pareto_k" if method == ISMethod.PSIS else "ess")
    if method == ISMethod.PSIS:
        data.append(good_k)
        index.append("good_k")
    data.append(n_data_points)
    index.append("subsample_size")
    return ELPDData(data=data, index=index)


def loo_from_matrix(log_likelihood, reff=1.0, scale=None, method="psis", pointwise=False, distributed=False, group=None):
    """LOO from an ``(n_obs, n_draws)`` log-likelihood matrix -- the engine-level entry point.

    ``log_likelihood`` may be a NumPy array (host) or a torch CUDA tensor (device-resident: no
    copies, the matrix is read exactly once).  With ``distributed=True`` (and
    ``torch.distributed`` initialised) every rank passes ITS OWN block of observations; the
    aggregates are merged with a single all-reduce and are identical on all ranks, pointwise
    outputs stay sharded.
    """
    method = parse_method(method)
    scale, scale_value = _scale_value(scale)
    n_local, n_samples = log_likelihood.shape
    good_k = min(1 - 1 / np.log10(n_samples), 0.7)
    res, agg = _engine_pass(log_likelihood, method, reff, scale_value, good_k, distributed, group)
    n_data_points = int(agg[0])
    warn = _diagnostic_warning(method, agg, good_k, n_samples)
    summ = _summaries(agg, n_data_points, scale_value)
    out = _pack(summ, n_samples, n_data_points, warn, scale, method, good_k, pointwise, res["loo_i"], res["diag"])
    out.method = method.value
    return out


def loo(data, pointwise=None, var_name=None, reff=None, scale=None, method="psis", moment_match=False,
        jacobian=None, mixture=False, **kwargs):
    """Leave-one-out cross-validation by importance sampling (PSIS by default).

    Same parameters, warnings, exceptions and ``ELPDData`` layout as ``pyloo.loo`` (loo.py:20-513).
    """
    idata = to_inference_data(data)
    log_likelihood = get_log_likelihood(idata, var_name=var_name)
    pointwise = rcParams["stats.ic_pointwise"] if pointwise is None else pointwise
    if jacobian is not None and not pointwise:
        raise ValueError(
            "Jacobian adjustment requires pointwise LOO results. "
            "Please set pointwise=True when using jacobian_adjustment."
        )
    matrix, obs_shape, obs_dims, coords = stack_samples(log_likelihood)  # loo.py:189
    n_samples = matrix.shape[-1]
    n_data_points = int(np.prod(obs_shape))  # loo.py:192
    scale, scale_value = _scale_value(scale)
    if reff is None:
        reff = _relative_efficiency(idata, n_samples)
    matrix = _replace_nan(matrix)  # loo.py:218-227
    method = _checked_method(method)  # loo.py:229-244
    good_k = min(1 - 1 / np.log10(n_samples), 0.7)  # loo.py:249
    if mixture:
        raise NotImplementedError("mixture=True (Mix-IS-LOO, loo.py:252-284) is outside the scope of pyloo_amd")
    if moment_match:
        if not pointwise:
            raise ValueError(
                "Moment matching requires pointwise LOO results. "
                "Please set pointwise=True when using moment_match=True."
            )
        raise NotImplementedError("moment_match=True (loo.py:441-511) needs the PyMC wrapper; outside the scope of pyloo_amd")

    res, agg = _engine_pass(matrix, method, reff, scale_value, good_k)
    warn = _diagnostic_warning(method, agg, good_k, n_samples)
    summ = _summaries(agg, n_data_points, scale_value)

    if not pointwise:
        out = _pack(summ, n_samples, n_data_points, warn, scale, method, good_k, False)
        out.method = method.value
        return out

    loo_i = np.asarray(res["loo_i"])
    if loo_i.size and np.allclose(loo_i, loo_i.flat[0]):  # loo.py:377-382
        warnings.warn(
            "The point-wise LOO is the same with the sum LOO, please double check "
            "the Observed RV in your model to make sure it returns element-wise logp.",
            stacklevel=2,
        )
    if jacobian is not None:  # loo.py:414-439
        jac = np.asarray(jacobian)
        if jac.shape != tuple(obs_shape):
            raise ValueError(f"Jacobian adjustment shape {jac.shape} does not match loo_i shape {tuple(obs_shape)}")
        loo_i = loo_i + jac.reshape(-1)
        elpd = loo_i.sum()
        se = (n_data_points * np.var(loo_i)) ** 0.5
        lppd = float(agg[AGG_SUM_LPPD])
        summ = {"elpd_loo": elpd, "se": se, "p_loo": lppd - elpd / scale_value,
                "p_loo_se": np.sqrt(np.sum(np.var(loo_i))), "looic": -2 * elpd, "looic_se": 2 * se}
    loo_da = wrap_obs(loo_i, obs_shape, obs_dims, coords, "loo_i")
    diag_da = wrap_obs(res["diag"], obs_shape, obs_dims, coords, "pareto_shape" if method == ISMethod.PSIS else "ess")
    out = _pack(summ, n_samples, n_data_points, warn, scale, method, good_k, True, loo_da, diag_da)
    out.method = method.value
    return out
