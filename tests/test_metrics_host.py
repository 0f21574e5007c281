"""``loo_predictive_metric`` / ``loo_score`` on the CPU: the oracle's closed forms and chains against the golden vectors made
from the reference's own primitives (tests/golden/make_golden_metrics.py), and the host Python of the two fronts (argument
handling, error texts, result objects) on top of the oracle-backed stand-in engine."""

import importlib

import numpy as np
import pytest

import pyloo_amd as pl
from conftest import load_golden
from fake_engine import OracleEngine
from oracle import psis_oracle as orc

CONT = ("mae", "mse", "rmse")
BIN = ("acc", "balanced_acc")


@pytest.fixture(scope="module")
def g():
    return load_golden("metrics")


@pytest.fixture(autouse=True)
def oracle_engine(monkeypatch):
    eng = OracleEngine()
    for name in ("pyloo_amd.base", "pyloo_amd.loo_predictive_metric", "pyloo_amd.loo_score"):
        monkeypatch.setattr(importlib.import_module(name), "get_engine", lambda device=None: eng)
    return eng


def groups(x, ll, y=None, chains=4, x2=None):
    """(n, S) matrices -> dict of groups with dims (chain, draw, obs); the stacked sample order is the column order."""
    def cdo(a):
        n, s = a.shape
        return np.ascontiguousarray(a.T.reshape(chains, s // chains, n))
    d = {"posterior_predictive": {"obs": cdo(x)}, "log_likelihood": {"obs": cdo(ll)}, "posterior": {"theta": np.zeros((1, x.shape[1]))}}
    if x2 is not None:
        d["predictions"] = {"obs": cdo(x2)}
    if y is not None:
        d["observed_data"] = {"obs": np.asarray(y)}
    return d


def pair(r):
    return np.array([r["estimate"], r["se"]])


def test_oracle_reducers_match_the_reference(g):
    for m in CONT:
        np.testing.assert_allclose(pair(orc.predictive_metric(g["red_y"], g["red_yhat"], m)), g[f"red_{m}"], rtol=1e-13)
    for m in BIN:
        np.testing.assert_allclose(pair(orc.predictive_metric(g["red_yb"], g["red_pb"], m)), g[f"red_{m}"], rtol=1e-13)


def test_oracle_chains_match_the_reference(g):
    reff = float(g["pm_reff"])
    for m in CONT:
        np.testing.assert_allclose(pair(orc.loo_predictive_metric_arrays(g["pm_x"], g["pm_ll"], g["pm_y"], m, reff)), g[f"pm_{m}"], rtol=1e-10)
    for m in BIN:
        np.testing.assert_allclose(pair(orc.loo_predictive_metric_arrays(g["pmb_x"], g["pmb_ll"], g["pmb_y"], m, 1.0)), g[f"pmb_{m}"], rtol=1e-10)
    for scale, tag in ((False, "crps"), (True, "scrps")):
        np.random.seed(1234)
        pw, k = orc.loo_score_arrays(g["pm_x"], g["sc_x2"], g["pm_y"], g["pm_ll"], reff, permutations=2, scale=scale)
        np.testing.assert_allclose(pw, g[f"sc_{tag}_pw"], rtol=1e-9)
        np.testing.assert_allclose(k, g["sc_k"], rtol=1e-9)


def test_front_reducers_match_the_reference(g):
    pm = importlib.import_module("pyloo_amd.loo_predictive_metric")

    for m in CONT:
        np.testing.assert_allclose(pair(pm._REDUCERS[m](g["red_y"], g["red_yhat"])), g[f"red_{m}"], rtol=1e-13)
    for m in BIN:
        np.testing.assert_allclose(pair(pm._REDUCERS[m](g["red_yb"], g["red_pb"])), g[f"red_{m}"], rtol=1e-13)
    with pytest.raises(ValueError, match="y and yhat must have the same length"):
        pm._mae(np.zeros(3), np.zeros(4))
    with pytest.raises(ValueError, match="y must contain values between 0 and 1"):
        pm._accuracy(np.array([0.0, 2.0]), np.array([0.1, 0.2]))
    with pytest.raises(ValueError, match="yhat must contain values between 0 and 1"):
        pm._balanced_accuracy(np.array([0.0, 1.0]), np.array([0.1, 1.2]))


def test_loo_predictive_metric_front(g):
    reff = float(g["pm_reff"])
    d = groups(g["pm_x"], g["pm_ll"])
    for m in CONT:
        np.testing.assert_allclose(pair(pl.loo_predictive_metric(d, g["pm_y"], metric=m, r_eff=reff)), g[f"pm_{m}"], rtol=1e-9)
        np.testing.assert_allclose(pair(pl.predictive_metric_from_matrix(g["pm_x"], g["pm_ll"], g["pm_y"], m, reff)), g[f"pm_{m}"], rtol=1e-9)
    db = groups(g["pmb_x"], g["pmb_ll"])
    for m in BIN:
        np.testing.assert_allclose(pair(pl.loo_predictive_metric(db, g["pmb_y"], var_name="obs", log_lik_var_name="obs", metric=m)),
                                   g[f"pmb_{m}"], rtol=1e-9)
    # error texts of loo_predictive_metric.py:153-206
    with pytest.raises(ValueError, match="does not have a nope group"):
        pl.loo_predictive_metric(d, g["pm_y"], group="nope")
    with pytest.raises(ValueError, match="does not have a nope group"):
        pl.loo_predictive_metric(d, g["pm_y"], log_lik_group="nope")
    with pytest.raises(ValueError, match="Variable 'zz' not found in log_likelihood group"):
        pl.loo_predictive_metric(d, g["pm_y"], log_lik_var_name="zz")
    with pytest.raises(ValueError, match=r"Length of y \(3\) must match the number of observations in x \(24\)"):
        pl.loo_predictive_metric(d, np.zeros(3))
    with pytest.raises(ValueError, match="Invalid metric: f1"):
        pl.loo_predictive_metric(d, g["pm_y"], metric="f1")
    two = dict(d)
    two["log_likelihood"] = {"a": d["log_likelihood"]["obs"], "b": d["log_likelihood"]["obs"]}
    with pytest.raises(ValueError, match="Multiple variables found in log_likelihood group. Please specify log_lik_var_name"):
        pl.loo_predictive_metric(two, g["pm_y"])


def test_loo_score_front(g):
    reff = float(g["pm_reff"])
    d = groups(g["pm_x"], g["pm_ll"], y=g["pm_y"], x2=g["sc_x2"])
    for scale, tag in ((False, "crps"), (True, "scrps")):
        np.random.seed(1234)
        res = pl.loo_score(d, x2_group="predictions", permutations=2, reff=reff, scale=scale, pointwise=True)
        np.testing.assert_allclose(res.pointwise, g[f"sc_{tag}_pw"], rtol=1e-9)
        np.testing.assert_allclose([res.estimates["Estimate"][0], res.estimates["SE"][0]], g[f"sc_{tag}_est"], rtol=1e-9)
        np.testing.assert_allclose(np.asarray(res.pareto_k), g["sc_k"], rtol=1e-9)
        assert res.good_k == min(1 - 1 / np.log10(g["pm_x"].shape[1]), 0.7) and res.warning == bool(np.any(g["sc_k"] > res.good_k))
    np.random.seed(7)
    plain = pl.loo_score(d, reff=reff, pointwise=False)  # x2 defaults to x itself (loo_score.py:478-486)
    assert plain.pareto_k is None and plain.warning is None and plain.pointwise.shape == (24,)
    np.random.seed(7)
    want, _ = orc.loo_score_arrays(g["pm_x"], g["pm_x"], g["pm_y"], g["pm_ll"], reff)
    np.testing.assert_allclose(plain.pointwise, want, rtol=1e-9)
    one_chain = groups(g["pm_x"], g["pm_ll"], y=g["pm_y"], chains=1)
    np.random.seed(7)
    assert np.isfinite(pl.loo_score(one_chain).estimates["Estimate"][0])  # reff=None, one chain: 1.0 (loo_score.py:208-209)
    # error texts of loo_score.py:349-414, 446-515
    with pytest.raises(ValueError, match="does not have a nope group"):
        pl.loo_score(d, x_group="nope", reff=1.0)
    with pytest.raises(ValueError, match="Variable 'zz' not found in observed_data group"):
        pl.loo_score(d, y_var="zz", reff=1.0)
    bad = dict(d)
    bad["predictions"] = {"obs": d["predictions"]["obs"][:, :, :5]}
    with pytest.raises(ValueError, match="x and x2 must have the same shape"):
        pl.loo_score(bad, x2_group="predictions", reff=1.0)
    bad_y = dict(d)
    bad_y["observed_data"] = {"obs": np.zeros(5)}
    with pytest.raises(ValueError, match="are not compatible with x dimensions"):
        pl.loo_score(bad_y, reff=1.0)
    nan_y = dict(d)
    nan_y["observed_data"] = {"obs": np.where(np.arange(24) == 3, np.nan, g["pm_y"])}
    with pytest.warns(UserWarning, match="NaN values detected"):
        pl.loo_score(nan_y, reff=1.0)
