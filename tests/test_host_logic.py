"""Host-side Python of pyloo_amd on CPU: argument handling, warnings, result packing.

The engine is replaced by tests/fake_engine.py (oracle-backed) so these tests follow the
reference's own test_loo.py / test_psis.py / test_base.py behaviours without a GPU."""

import warnings

import numpy as np
import pytest

import pyloo_amd as pl
from conftest import load_golden
from fake_engine import OracleEngine
from oracle import psis_oracle as orc


@pytest.fixture(autouse=True)
def oracle_engine(monkeypatch):
    eng = OracleEngine()
    import importlib

    for name in ("pyloo_amd.base", "pyloo_amd.loo"):  # (pyloo_amd.loo the attribute is the function)
        monkeypatch.setattr(importlib.import_module(name), "get_engine", lambda device=None: eng)
    return eng


def idata(ll_matrix, chains=4, posterior=True):
    """(N, S) matrix -> dict InferenceData stand-in with dims (chain, draw, obs)."""
    n, s = ll_matrix.shape
    arr = np.moveaxis(ll_matrix.reshape(n, chains, s // chains), 0, -1)
    d = {"log_likelihood": {"obs": arr}}
    if posterior:
        d["posterior"] = {"mu": np.zeros((chains, s // chains))}
    return d


@pytest.fixture(scope="module")
def ll8():
    rng = np.random.default_rng(44)  # conftest.py:106-109 of the reference uses seed 44
    return -0.3 * rng.exponential(size=(8, 2000)) - 3.0


def test_loo_result_layout_and_values(ll8):
    res = pl.loo(idata(ll8), reff=0.7)
    assert list(res.index) == ["elpd_loo", "se", "p_loo", "p_loo_se", "n_samples", "n_data_points", "warning",
                               "scale", "looic", "looic_se", "good_k", "subsample_size"]  # loo.py:344-367
    want = orc.loo_arrays(ll8, 0.7)
    for key in ("elpd_loo", "se", "p_loo", "p_loo_se", "looic", "looic_se"):
        np.testing.assert_allclose(res[key], want[key], rtol=1e-12)
    assert res["n_samples"] == 2000 and res["n_data_points"] == 8 and res["subsample_size"] == 8
    assert res["scale"] == "log" and res["good_k"] == pytest.approx(min(1 - 1 / np.log10(2000), 0.7))
    pw = pl.loo(idata(ll8), reff=0.7, pointwise=True)
    assert list(pw.index) == ["elpd_loo", "se", "p_loo", "p_loo_se", "n_samples", "n_data_points", "warning", "loo_i",
                              "scale", "looic", "looic_se", "pareto_k", "good_k", "subsample_size"]  # loo.py:384-412
    np.testing.assert_allclose(np.asarray(pw["loo_i"]), want["loo_i"], rtol=1e-12)
    np.testing.assert_allclose(np.asarray(pw["pareto_k"]), want["khat"], rtol=1e-12)


@pytest.mark.parametrize("scale,value", [("log", 1), ("negative_log", -1), ("deviance", -2)])
def test_scales(ll8, scale, value):  # test_loo.py:32-40
    res = pl.loo(idata(ll8), reff=1.0, scale=scale)
    want = orc.loo_arrays(ll8, 1.0, value)
    for key in ("elpd_loo", "se", "p_loo", "looic", "looic_se"):
        np.testing.assert_allclose(res[key], want[key], rtol=1e-12)
    assert res["scale"] == scale


def test_bad_arguments(ll8):
    with pytest.raises(TypeError, match='Valid scale values are "deviance", "log", "negative_log"'):
        pl.loo(idata(ll8), reff=1.0, scale="invalid")  # test_loo.py:64-68
    with pytest.raises(TypeError):
        pl.loo({"posterior": {"mu": np.zeros((4, 100))}})  # test_loo.py:71-74
    with pytest.raises(TypeError, match="Must be able to extract a posterior group from data"):
        pl.loo(idata(ll8, posterior=False), reff=None)  # test_loo.py:77-86
    assert pl.loo(idata(ll8, posterior=False), reff=0.7) is not None
    with pytest.raises(ValueError, match="Invalid method 'invalid'"):
        pl.loo(idata(ll8), reff=1.0, method="invalid")  # test_loo.py:227-229
    with pytest.raises(ValueError, match="Jacobian adjustment requires pointwise"):
        pl.loo(idata(ll8), reff=1.0, jacobian=np.zeros(8))  # loo.py:183-187
    with pytest.raises(ValueError, match="does not match loo_i shape"):
        pl.loo(idata(ll8), reff=1.0, pointwise=True, jacobian=np.zeros(7))
    two = idata(ll8)
    two["log_likelihood"]["obs2"] = two["log_likelihood"]["obs"]
    with pytest.raises(TypeError, match="several log likelihood arrays"):
        pl.loo(two, reff=1.0)  # test_loo.py:190-198
    assert pl.loo(two, reff=1.0, var_name="obs") is not None
    with pytest.raises(TypeError, match="No log likelihood data named nope"):
        pl.loo(two, reff=1.0, var_name="nope")
    with pytest.raises(NotImplementedError):
        pl.loo(idata(ll8), reff=1.0, mixture=True)
    with pytest.raises(ValueError, match="Moment matching requires pointwise"):
        pl.loo(idata(ll8), reff=1.0, moment_match=True)


def test_one_chain_needs_no_reff(ll8):  # loo.py:209-210
    res = pl.loo(idata(ll8, chains=1))
    np.testing.assert_allclose(res["elpd_loo"], orc.loo_arrays(ll8, 1.0)["elpd_loo"], rtol=1e-12)


def test_warnings(ll8):
    bad = ll8.copy()
    bad[1] = -3.0 * np.random.default_rng(1).exponential(size=2000)  # very heavy tail -> khat > 0.7
    with pytest.warns(UserWarning, match="Estimated shape parameter of Pareto distribution is greater than 0.70 for 1 observations"):
        res = pl.loo(idata(bad), reff=1.0, pointwise=True)  # test_loo.py:89-97
    assert res["warning"] and np.any(np.asarray(res["pareto_k"]) > res["good_k"])
    const = np.ones_like(ll8)
    with pytest.warns(UserWarning) as rec:
        pl.loo(idata(const), reff=1.0, pointwise=True)  # test_loo.py:100-108
    assert any("The point-wise LOO is the same" in str(w.message) for w in rec)
    nan = ll8.copy()
    nan[0, 0] = np.nan
    with pytest.warns(UserWarning, match="NaN values detected"):
        res = pl.loo(idata(nan), reff=1.0)  # test_loo.py:139-153
    assert not np.isnan(res["elpd_loo"])
    ext = ll8.copy()
    ext[0, 0], ext[1, 0] = 1e10, -1e10
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.isfinite(pl.loo(idata(ext), reff=1.0)["elpd_loo"])  # test_loo.py:156-171


def test_methods(ll8):  # test_loo.py:201-245
    with pytest.warns(UserWarning, match="Using SIS for LOO computation"):
        sis = pl.loo(idata(ll8), reff=1.0, pointwise=True, method="sis")
    assert "ess" in sis and "pareto_k" not in sis and "good_k" not in sis
    with pytest.warns(UserWarning, match="Using TIS for LOO computation"):
        tis = pl.loo(idata(ll8), reff=1.0, pointwise=True, method=pl.ISMethod.TIS)
    assert "ess" in tis and np.all(np.asarray(tis["ess"]) >= 1) and np.all(np.asarray(tis["ess"]) <= 2000)
    with pytest.warns(UserWarning, match="Low effective sample size detected"):
        res = pl.loo(idata(ll8 * 40), reff=1.0, method="sis")  # test_loo.py:232-243
    assert res["warning"]
    assert "good_k" in pl.loo(idata(ll8), reff=1.0)


def test_jacobian(ll8):  # test_loo.py:307-336 / loo.py:414-439
    jac = np.linspace(-0.5, 0.5, 8)
    base = pl.loo(idata(ll8), reff=1.0, pointwise=True)
    adj = pl.loo(idata(ll8), reff=1.0, pointwise=True, jacobian=jac)
    np.testing.assert_allclose(np.asarray(adj["loo_i"]), np.asarray(base["loo_i"]) + jac, rtol=1e-13)
    np.testing.assert_allclose(adj["elpd_loo"], base["elpd_loo"] + jac.sum(), rtol=1e-12)
    np.testing.assert_allclose(adj["looic"], -2 * adj["elpd_loo"], rtol=1e-13)


def test_multidim_observations():
    rng = np.random.default_rng(0)
    arr = rng.normal(size=(4, 100, 10, 2))  # test_loo.py:22-29
    res = pl.loo({"log_likelihood": {"obs": arr}}, reff=1.0, pointwise=True)
    assert res["n_data_points"] == 20 and res["n_samples"] == 400
    assert np.asarray(res["loo_i"]).shape == (10, 2) and np.asarray(res["pareto_k"]).shape == (10, 2)
    mat = np.moveaxis(arr.reshape(400, 10, 2), 0, -1).reshape(20, 400)
    np.testing.assert_allclose(res["elpd_loo"], orc.loo_arrays(mat, 1.0)["elpd_loo"], rtol=1e-12)


def test_psislw_and_compute_importance_weights_fronts():
    g = load_golden("shapes")
    lw, k = pl.psislw(g["x1"], 0.7)  # test_psis.py:49-58: 1-D input -> 0-d ndarray k
    assert isinstance(k, np.ndarray) and k.shape == () and lw.shape == g["x1"].shape
    np.testing.assert_allclose(lw, g["lw1"], rtol=1e-12)
    lw, k = pl.psislw(g["x3"], 0.7)
    assert lw.shape == (2, 3, 100) and k.shape == (2, 3)
    np.testing.assert_allclose(k, g["k3"], rtol=1e-12)
    keep = g["x3"].copy()
    pl.compute_importance_weights(g["x3"], "psis", 0.7)
    assert np.array_equal(keep, g["x3"])  # inputs are never modified (base.py:112)
    lw2, ess = pl.compute_importance_weights(g["x3"], method="SIS")
    assert ess.shape == (2, 3) and np.allclose(np.exp(lw2).sum(-1), 1.0)
    with pytest.raises(ValueError, match="Invalid method 'nope'. Must be one of: psis, sis, tis"):
        pl.compute_importance_weights(g["x3"], "nope")  # base.py:100-107
    with pytest.raises(ValueError, match="log_weights must be provided"):
        pl.compute_importance_weights(None)  # base.py:109-110
    with pytest.raises(IndexError):
        pl.psislw(np.zeros(1))  # x_sort_ind[-2] of a length-1 vector (psis.py:136)
    with pytest.raises(ZeroDivisionError):
        pl.psislw(g["x1"], reff=0.0)  # same Python expression as psis.py:89
    lw, k = pl.psislw(np.zeros((0, 50)))
    assert lw.shape == (0, 50) and k.shape == (0,)


def test_elpd_report_format(ll8):
    res = pl.loo(idata(ll8), reff=1.0)
    text = str(res)
    assert text.startswith("\nComputed from 2000 posterior samples and 8 observations log-likelihood matrix.\n")
    assert "         Estimate       SE\nelpd_loo   " in text and "\np_loo       " in text and "\nlooic      " in text
    assert "All Pareto k estimates are good (k < 0.7).\nSee help('pareto-k-diagnostic') for details." in text  # README.md:76-84
    line = [l for l in text.split("\n") if l.startswith("elpd_loo")][0]
    assert line == f"elpd_loo   {res['elpd_loo']:<8.2f}    {res['se']:<.2f}"
    pw = pl.loo(idata(ll8), reff=1.0, pointwise=True)
    assert "All Pareto k estimates are good" in str(pw)
    bad = ll8.copy()
    bad[1] = -3.0 * np.random.default_rng(1).exponential(size=2000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rep = str(pl.loo(idata(bad), reff=1.0, pointwise=True))
        short = str(pl.loo(idata(bad), reff=1.0))
    assert "Pareto k diagnostic values:" in rep and "(good)" in rep and "(very bad)" in rep
    assert "There has been a warning during the calculation" in rep
    assert "Some Pareto k diagnostic values are high" in short
    cp = pw.copy()
    assert isinstance(cp, pl.ELPDData) and cp["elpd_loo"] == pw["elpd_loo"]


def test_rcparams():
    assert pl.rcParams["stats.ic_pointwise"] is False and pl.rcParams["stats.ic_scale"] == "log"
    with pytest.raises(ValueError, match="Key stats.ic_scale"):
        pl.rcParams["stats.ic_scale"] = "bogus"
    with pytest.raises(KeyError):
        pl.rcParams["nope"] = 1
    with pytest.raises(TypeError):
        del pl.rcParams["stats.ic_scale"]
    pl.rcParams["stats.ic_scale"] = "Deviance"
    try:
        assert pl.rcParams["stats.ic_scale"] == "deviance"
    finally:
        pl.rcParams["stats.ic_scale"] = "log"
    assert list(pl.rcParams) == sorted(pl.rcParams.keys())


def _loo_i_reference(ll_row, reff, scale_value=1, method="psis"):
    """loo_i.py:183-239 restated with the oracle (weights of one row, SE from the weights)."""
    S = ll_row.shape[0]
    lw, diag = orc.importance_weights(-ll_row[None, :], method, reff)
    lwll = lw + ll_row[None, :]
    loo = scale_value * orc.lse(lwll[0])
    w = np.exp(lwll - lwll.max(axis=-1, keepdims=True))
    w /= w.sum(axis=-1, keepdims=True)
    e_epd = np.exp(loo)
    var = np.sum(w**2 * (np.exp(ll_row[None, :]) - e_epd) ** 2) / reff
    se = np.sqrt(np.log1p(var / e_epd**2))
    lppd = orc.lse(ll_row, b_inv=S)
    return loo, se, lppd - loo / scale_value, diag


def test_loo_i(ll8, monkeypatch):  # test_loo_i.py of the reference: layout, values, errors
    import importlib

    monkeypatch.setattr(importlib.import_module("pyloo_amd.base"), "get_engine", lambda device=None: OracleEngine())
    d = idata(ll8)
    res = pl.loo_i(3, d, reff=0.8)
    assert list(res.index) == ["elpd_loo", "se", "p_loo", "n_samples", "n_data_points", "warning", "scale", "good_k"]  # loo_i.py:242-258
    loo, se, p_loo, diag = _loo_i_reference(ll8[3], 0.8)
    np.testing.assert_allclose(res["elpd_loo"], loo, rtol=1e-12)
    np.testing.assert_allclose(res["se"], se, rtol=1e-10)
    np.testing.assert_allclose(res["p_loo"], p_loo, rtol=1e-10)
    assert res["n_data_points"] == 1 and res["n_samples"] == 2000
    pw = pl.loo_i(3, d, reff=0.8, pointwise=True, scale="deviance")
    assert list(pw.index) == ["elpd_loo", "se", "p_loo", "n_samples", "n_data_points", "warning", "loo_i", "scale",
                              "pareto_k", "good_k"]  # loo_i.py:260-292
    np.testing.assert_allclose(pw["elpd_loo"], -2 * loo, rtol=1e-12)
    np.testing.assert_allclose(np.asarray(pw["pareto_k"]).ravel(), diag, rtol=1e-12)
    with pytest.warns(UserWarning, match="Using SIS"):
        sis = pl.loo_i(0, d, reff=1.0, method="sis", pointwise=True)
    assert "ess" in sis.index and "good_k" not in sis.index
    with pytest.raises(ValueError, match="single integer"):
        pl.loo_i([0, 1], d, reff=1.0)
    with pytest.raises(TypeError, match="must be an integer"):
        pl.loo_i("a", d, reff=1.0)
    with pytest.raises(IndexError, match="out of bounds"):
        pl.loo_i(8, d, reff=1.0)
    with pytest.raises(ValueError, match="Invalid method"):
        pl.loo_i(0, d, reff=1.0, method="nope")
    with pytest.raises(TypeError, match="Valid scale values"):
        pl.loo_i(0, d, reff=1.0, scale="nope")


def test_loo_subsample_posterior_correction(monkeypatch, oracle_engine):
    """loo_subsample.py:333-370 through the front: ``log_p`` / ``log_q`` re-draw the draws of the sampled rows ("psis": a
    weighted permutation -- the estimates stay what they are without it), unequal lengths are refused, and a resampling that
    fails (non-finite ratios: the reference's own function ends in an exception there) falls back with the reference's warning."""
    import importlib

    ls = importlib.import_module("pyloo_amd.loo_subsample")
    monkeypatch.setattr(ls, "get_engine", lambda device=None: oracle_engine)
    rng = np.random.default_rng(3)
    ll = -0.4 * rng.exponential(size=(60, 2000)) - 2.0
    log_q = rng.normal(size=2000)
    log_p = log_q + 0.5 * rng.normal(size=2000)
    obs = np.arange(0, 60, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = pl.loo_subsample(idata(ll), observations=obs, loo_approximation="lpd", reff=1.0)
        corrected = pl.loo_subsample(idata(ll), observations=obs, loo_approximation="lpd", reff=1.0, log_p=log_p, log_q=log_q, seed=5)
    np.testing.assert_allclose(corrected["elpd_loo"], plain["elpd_loo"], rtol=1e-10)
    np.testing.assert_allclose(corrected["p_loo"], plain["p_loo"], rtol=1e-8)
    with pytest.raises(ValueError, match="log_p and log_q must have the same length, got 2000 and 1999"):
        pl.loo_subsample(idata(ll), observations=obs, loo_approximation="lpd", reff=1.0, log_p=log_p, log_q=log_q[:-1])
    bad = log_p.copy()
    bad[[4, 9]] = [np.inf, np.nan]
    with pytest.warns(UserWarning, match="Importance resampling failed: .* Falling back to original samples."):
        fell_back = pl.loo_subsample(idata(ll), observations=obs, loo_approximation="lpd", reff=1.0, log_p=bad, log_q=log_q, seed=5)
    np.testing.assert_allclose(fell_back["elpd_loo"], plain["elpd_loo"], rtol=1e-12)
    # log_p / log_q of another length than the draws: the reference's reshape to the stacked shape raises
    # (loo_subsample.py:348-356) and the same fallback takes over -- shorter and longer
    for n_other in (1500, 2500):
        lq = rng.normal(size=n_other)
        lp = lq + 0.5 * rng.normal(size=n_other)
        with pytest.warns(UserWarning, match="Importance resampling failed: cannot reshape array of size .* Falling back to original samples."):
            other = pl.loo_subsample(idata(ll), observations=obs, loo_approximation="lpd", reff=1.0, log_p=lp, log_q=lq, seed=5)
        np.testing.assert_allclose(other["elpd_loo"], plain["elpd_loo"], rtol=1e-12)
