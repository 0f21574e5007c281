"""The ArviZ / xarray surface of the drop-in (north star: "keeping the pl.loo()/pl.psislw() API and ArviZ InferenceData
surface"; reference loo.py:179-193, utils.py:21-79,257-302, base.py:93-98,168-173, e_loo.py:198-212).

Two layers:
  * with real ArviZ installed (``importorskip``): C1 -- ``centered_eight`` through ``pl.loo`` against the numbers the
    reference prints in its README (README.md:76-81), and ``pl.psislw`` on the stacked DataArray (GPU marker: the engine runs);
  * without it (this container, the GPU box): the DataArray branches are driven on the CPU with ``tests/fake_xarray.py`` patched
    in as ``xr`` and the oracle-backed stand-in engine, so that dim handling, coordinate carry-over and output wrapping
    execute and are checked against the plain-array path."""

import importlib
import warnings

import numpy as np
import pytest

import fake_xarray
from fake_engine import OracleEngine
from oracle import psis_oracle as orc


@pytest.fixture()
def patched(monkeypatch):
    """pyloo_amd with ``xr`` = the minimal DataArray module and the oracle engine behind every front."""
    import pyloo_amd as pl

    eng = OracleEngine()
    for name in ("pyloo_amd.utils", "pyloo_amd.base", "pyloo_amd.e_loo"):
        monkeypatch.setattr(importlib.import_module(name), "xr", fake_xarray)
    for name in ("pyloo_amd.loo", "pyloo_amd.base", "pyloo_amd.e_loo"):
        monkeypatch.setattr(importlib.import_module(name), "get_engine", lambda device=None: eng)
    return pl


def _ll(rng, chains=4, draws=150, shape=(3, 5)):
    k = rng.uniform(0.1, 0.9, size=shape)
    return -k * rng.exponential(size=(chains, draws) + shape) - 1.5


def test_stack_samples_dataarray_branch(patched):
    """loo.py:189: ``.stack(__sample__=("chain", "draw"))`` of a DataArray whose dims are NOT in (chain, draw, *obs) order."""
    from pyloo_amd.utils import stack_samples, wrap_obs

    rng = np.random.default_rng(1)
    raw = _ll(rng)  # (chain, draw, school, year)
    da = fake_xarray.DataArray(np.moveaxis(raw, (0, 1), (2, 3)), dims=("school", "year", "chain", "draw"),
                               coords={"school": list("abc"), "year": np.arange(2001, 2006)}, name="obs")
    m, obs_shape, obs_dims, coords = stack_samples(da)
    want, shape2, dims2, _ = stack_samples(raw)  # the plain-array path on (chain, draw, *obs)
    assert obs_shape == shape2 == (3, 5) and obs_dims == ("school", "year") and m.shape == (15, 600)
    np.testing.assert_array_equal(np.asarray(m), np.asarray(want))  # chain-major draws, observations in C order
    assert list(coords["school"]) == list("abc") and coords["year"][0] == 2001
    back = wrap_obs(np.arange(15.0), obs_shape, obs_dims, coords, "loo_i")
    assert isinstance(back, fake_xarray.DataArray) and back.dims == ("school", "year") and back.name == "loo_i"
    assert back.values[2, 4] == 14.0 and list(back.coords["school"]) == list("abc")
    already = stack_samples(da.stack(__sample__=("chain", "draw")))[0]  # an already stacked array is taken as it is
    np.testing.assert_array_equal(np.asarray(already), np.asarray(want))


def test_loo_through_dataarray_log_likelihood(patched):
    """``pl.loo`` on an InferenceData-like object whose log likelihood is a DataArray: numbers equal the ndarray path, the
    pointwise outputs come back as DataArrays on the observation dims with their coordinates (loo.py:189, 607)."""
    pl = patched
    rng = np.random.default_rng(2)
    raw = _ll(rng)

    class IData:
        log_likelihood = {"obs": fake_xarray.DataArray(raw, dims=("chain", "draw", "school", "year"),
                                                       coords={"school": list("abc")}, name="obs")}
        posterior = {"mu": rng.normal(size=(4, 150))}

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = pl.loo(IData(), pointwise=True, reff=0.9)
        want = pl.loo({"log_likelihood": {"obs": raw}, "posterior": {"mu": IData.posterior["mu"]}}, pointwise=True, reff=0.9)
    for key in ("elpd_loo", "se", "p_loo", "p_loo_se", "looic", "looic_se"):
        assert got[key] == want[key], key
    assert isinstance(got["loo_i"], fake_xarray.DataArray) and got["loo_i"].dims == ("school", "year")
    assert got["pareto_k"].name == "pareto_shape" and list(got["loo_i"].coords["school"]) == list("abc")
    np.testing.assert_array_equal(got["loo_i"].values, np.asarray(want["loo_i"]))
    ref = orc.loo_arrays(np.moveaxis(raw.reshape(600, 15), 0, -1), 0.9)
    np.testing.assert_allclose(got["elpd_loo"], ref["elpd_loo"], rtol=1e-12)


def test_psislw_dataarray_branches(patched):
    """base.py:93-98,168-173 / psis.py:104-111: chain/draw are stacked when ``__sample__`` is missing, the weights come back
    with the INPUT's dim order, the diagnostic on the observation dims; no sample dims at all is a ValueError."""
    pl = patched
    rng = np.random.default_rng(3)
    raw = -_ll(rng, shape=(6,))  # log ratios (chain, draw, obs)
    da = fake_xarray.DataArray(raw, dims=("chain", "draw", "obs"), coords={"obs": np.arange(10, 16)})
    lw, k = pl.psislw(da, reff=1.0)
    assert lw.dims == ("obs", "__sample__") and k.dims == ("obs",) and k.name == "pareto_shape" and lw.name == "log_weights"
    flat = np.moveaxis(raw.reshape(600, 6), 0, -1)
    want_lw, want_k = pl.psislw(flat, reff=1.0)
    np.testing.assert_array_equal(lw.values, want_lw)
    np.testing.assert_array_equal(k.values, want_k)
    assert list(k.coords["obs"]) == list(range(10, 16))
    # sample dim first: the weights keep that order
    first = fake_xarray.DataArray(flat.T, dims=("__sample__", "obs"))
    lw2, k2 = pl.compute_importance_weights(first, method="psis", reff=1.0)
    assert lw2.dims == ("__sample__", "obs")
    np.testing.assert_array_equal(lw2.values, want_lw.T)
    np.testing.assert_array_equal(k2.values, want_k)
    # SIS: the diagnostic is the ESS
    _, ess = pl.compute_importance_weights(da, method="sis")
    assert ess.name == "ess" and ess.dims == ("obs",)
    with pytest.raises(ValueError, match="__sample__"):
        pl.compute_importance_weights(fake_xarray.DataArray(flat, dims=("obs", "time")))


def test_e_loo_pairs_dataarrays_by_dimension_name(patched):
    """e_loo.py:198-212: data and weights pair up by dimension name, not by position; other names are an error."""
    pl = patched
    rng = np.random.default_rng(4)
    x = rng.normal(size=(4, 100, 3, 2))
    lr = 0.4 * rng.exponential(size=(4, 100, 3, 2))
    xd = fake_xarray.DataArray(x, dims=("chain", "draw", "a", "b"), coords={"a": [1, 2, 3]})
    lw_plain, _ = pl.psislw(np.moveaxis(lr.reshape(400, 3, 2), 0, -1))
    # the weights with their observation dims in the OTHER order
    lwd = fake_xarray.DataArray(np.swapaxes(lw_plain, 0, 1), dims=("b", "a", "__sample__"))
    got = pl.e_loo(xd, log_weights=lwd, type="mean")
    want = pl.e_loo(np.moveaxis(x.reshape(400, 3, 2), 0, -1), log_weights=lw_plain, type="mean")
    np.testing.assert_allclose(np.asarray(getattr(got.value, "values", got.value)), np.asarray(want.value), rtol=1e-13)
    assert got.value.dims == ("a", "b") and list(got.value.coords["a"]) == [1, 2, 3]
    with pytest.raises(ValueError, match="same names"):
        pl.e_loo(xd, log_weights=fake_xarray.DataArray(lw_plain, dims=("a", "c", "__sample__")), type="mean")
    # weights instead of log-weights (e_loo.py:202-203) as a DataArray
    wd = fake_xarray.DataArray(np.exp(lw_plain), dims=("a", "b", "__sample__"))
    got2 = pl.e_loo(xd, weights=wd, type="mean")
    np.testing.assert_allclose(np.asarray(got2.value.values), np.asarray(want.value), rtol=1e-12)


# ---------------------------------------------------------------------------------------------------------------------------
# with the real packages (skipped where they are not installed: the build container and the GPU box have neither)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_c1_centered_eight_through_arviz():
    """BASELINE.json config 1: ``pl.loo(az.load_arviz_data("centered_eight"))`` against the reference's README numbers
    (README.md:76-81: elpd_loo -30.78, SE 1.35, p_loo 0.95, p_loo_se 0.48, looic 61.56, looic SE 2.69; 2000 samples, 8 points)."""
    az = pytest.importorskip("arviz", reason="C1 needs ArviZ (absent in the build container and on the GPU box)")
    import pyloo_amd as pl

    idata = az.load_arviz_data("centered_eight")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = pl.loo(idata, pointwise=True)
    assert res["n_samples"] == 2000 and res["n_data_points"] == 8
    for key, want in (("elpd_loo", -30.78), ("se", 1.35), ("p_loo", 0.95), ("p_loo_se", 0.48), ("looic", 61.56), ("looic_se", 2.69)):
        assert round(float(res[key]), 2) == want, (key, float(res[key]))
    assert res["loo_i"].dims == ("school",)
    stacked = (-idata.log_likelihood["obs"]).stack(__sample__=("chain", "draw"))
    lw, k = pl.psislw(stacked, reff=1.0)
    assert lw.dims == stacked.dims and k.dims == ("school",)


def test_real_xarray_branches_when_installed():
    """The same DataArray branches against real xarray, where it is installed (CPU, oracle engine)."""
    xr = pytest.importorskip("xarray", reason="xarray is absent in the build container and on the GPU box")
    from pyloo_amd.utils import stack_samples

    rng = np.random.default_rng(5)
    raw = _ll(rng)
    da = xr.DataArray(raw, dims=("chain", "draw", "school", "year"))
    m, obs_shape, obs_dims, _ = stack_samples(da)
    want = stack_samples(raw)[0]
    assert obs_shape == (3, 5) and obs_dims == ("school", "year")
    np.testing.assert_array_equal(np.asarray(m), np.asarray(want))
