"""waic() front on CPU (engine replaced by the oracle-backed stand-in): layout, scales, warnings, errors --
the behaviours the reference's own test_waic.py exercises (tests/base_tests/test_waic.py:16-110), plus the
oracle's WAIC arithmetic against independently written NumPy."""

import importlib
import warnings

import numpy as np
import pytest

import pyloo_amd as pl
from fake_engine import OracleEngine
from oracle import psis_oracle as orc


@pytest.fixture(autouse=True)
def oracle_engine(monkeypatch):
    eng = OracleEngine()
    monkeypatch.setattr(importlib.import_module("pyloo_amd.waic"), "get_engine", lambda device=None: eng)
    return eng


def idata(ll_matrix, chains=4):
    n, s = ll_matrix.shape
    arr = np.moveaxis(ll_matrix.reshape(n, chains, s // chains), 0, -1)
    return {"log_likelihood": {"obs": arr}, "posterior": {"mu": np.zeros((chains, s // chains))}}


@pytest.fixture(scope="module")
def ll8():
    rng = np.random.default_rng(44)
    return -0.3 * rng.exponential(size=(8, 2000)) - 3.0


def test_oracle_waic_arithmetic(ll8):
    """waic.py:137-161 restated twice: the oracle against plain scipy/numpy one-liners."""
    from scipy.special import logsumexp

    w = orc.waic_arrays(ll8, 1)
    lppd = logsumexp(ll8, axis=1) - np.log(ll8.shape[1])
    var = np.var(ll8, axis=1)
    np.testing.assert_allclose(w["lppd_i"], lppd, rtol=1e-13)
    np.testing.assert_allclose(w["var_i"], var, rtol=1e-13)
    np.testing.assert_allclose(w["elpd_waic"], np.sum(lppd - var), rtol=1e-13)
    np.testing.assert_allclose(w["se"], np.sqrt(8 * np.var(lppd - var)), rtol=1e-12)
    np.testing.assert_allclose(w["p_waic"], var.sum(), rtol=1e-13)


def test_layout_and_values(ll8):
    res = pl.waic(idata(ll8))
    assert list(res.index) == ["elpd_waic", "se", "p_waic", "n_samples", "n_data_points", "warning", "scale"]  # waic.py:164-176
    want = orc.waic_arrays(ll8, 1)
    for key in ("elpd_waic", "se", "p_waic"):
        np.testing.assert_allclose(res[key], want[key], rtol=1e-12)
    assert res["n_samples"] == 2000 and res["n_data_points"] == 8 and res["scale"] == "log"
    assert res["warning"] == want["warning"]
    pw = pl.waic(idata(ll8), pointwise=True)
    assert list(pw.index) == ["elpd_waic", "se", "p_waic", "n_samples", "n_data_points", "warning", "waic_i", "scale"]  # waic.py:188-207
    np.testing.assert_allclose(np.asarray(pw["waic_i"]), want["waic_i"], rtol=1e-12)
    assert "elpd_waic" in str(res)


@pytest.mark.parametrize("scale,value", [("log", 1), ("negative_log", -1), ("deviance", -2)])
def test_scales(ll8, scale, value):  # test_waic.py:38-44
    res = pl.waic(idata(ll8), scale=scale)
    np.testing.assert_allclose(res["elpd_waic"], value * orc.waic_arrays(ll8, 1)["elpd_waic"], rtol=1e-12)
    assert res["scale"] == scale


def test_invalid_scale_and_missing_loglik(ll8):  # test_waic.py:47-49, 73-76
    with pytest.raises(TypeError, match="Valid scale values are"):
        pl.waic(idata(ll8), scale="invalid")
    with pytest.raises(TypeError):
        pl.waic({"posterior": {"mu": np.zeros((4, 10))}})


def test_nan_inf_warnings(ll8):  # test_waic.py:52-60
    bad = ll8.copy()
    bad[0, :] = np.nan
    bad[1, :] = np.inf
    with pytest.warns(UserWarning, match="NaN values detected"):
        with pytest.warns(UserWarning, match="Infinite values detected"):
            res = pl.waic(idata(bad))
    want = orc.waic_arrays(bad, 1)
    np.testing.assert_allclose(res["elpd_waic"], want["elpd_waic"], rtol=1e-12)


def test_variance_warning_and_constant(ll8):
    wide = ll8 * 10.0  # variance over draws far above 0.4
    with pytest.warns(UserWarning, match="exceeds 0.4"):
        res = pl.waic(idata(wide))
    assert bool(res["warning"]) is True
    const = np.full((5, 400), -1.5)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = pl.waic(idata(const), pointwise=True)
    assert any("point-wise WAIC is the same" in str(w.message) for w in rec)  # waic.py:178-184
    np.testing.assert_allclose(out["p_waic"], 0.0, atol=1e-12)


def test_several_log_likelihoods(ll8):  # test_waic.py:63-70
    d = idata(ll8)
    d["log_likelihood"]["obs2"] = d["log_likelihood"]["obs"]
    with pytest.raises(TypeError, match="Found several log likelihood arrays"):
        pl.waic(d)
    assert pl.waic(d, var_name="obs") is not None
