"""``loo_predictive_metric`` / ``loo_score`` through the real engine (``-m gpu``): against the golden vectors made from the
reference's primitives (tests/golden/make_golden_metrics.py) and, on larger seeded inputs and CUDA tensors, the oracle."""

import numpy as np
import pytest

import pyloo_amd as pl
from conftest import load_golden
from oracle import psis_oracle as orc
from test_metrics_host import BIN, CONT, groups, pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def oracle_engine():  # (shadows the CPU suite's autouse stand-in: these tests run on the real engine)
    return None


@pytest.fixture(scope="module")
def g():
    return load_golden("metrics")


def test_golden_predictive_metrics(g):
    reff = float(g["pm_reff"])
    d = groups(g["pm_x"], g["pm_ll"])
    for m in CONT:
        np.testing.assert_allclose(pair(pl.loo_predictive_metric(d, g["pm_y"], metric=m, r_eff=reff)), g[f"pm_{m}"], rtol=1e-9)
    db = groups(g["pmb_x"], g["pmb_ll"])
    for m in BIN:
        np.testing.assert_allclose(pair(pl.loo_predictive_metric(db, g["pmb_y"], metric=m)), g[f"pmb_{m}"], rtol=1e-9)


def test_golden_scores(g):
    reff = float(g["pm_reff"])
    d = groups(g["pm_x"], g["pm_ll"], y=g["pm_y"], x2=g["sc_x2"])
    for scale, tag in ((False, "crps"), (True, "scrps")):
        np.random.seed(1234)
        res = pl.loo_score(d, x2_group="predictions", permutations=2, reff=reff, scale=scale, pointwise=True)
        np.testing.assert_allclose(res.pointwise, g[f"sc_{tag}_pw"], rtol=1e-9)
        np.testing.assert_allclose([res.estimates["Estimate"][0], res.estimates["SE"][0]], g[f"sc_{tag}_est"], rtol=1e-9)
        np.testing.assert_allclose(np.asarray(res.pareto_k), g["sc_k"], rtol=1e-9)


def test_device_matrices_vs_oracle():
    """(n_obs, n_draws) CUDA tensors: nothing but the n predictions / scores leaves the device."""
    import torch

    rng = np.random.default_rng(99)
    n, s = 96, 4000
    theta = rng.normal(size=(1, s)) * 0.4
    y = rng.normal(size=n)
    ll = -0.5 * (y[:, None] - theta) ** 2 * rng.uniform(0.5, 2.0, size=(n, 1))
    x = theta + rng.normal(size=(n, s))
    tx, tl, ty = torch.from_numpy(x).cuda(), torch.from_numpy(ll).cuda(), torch.from_numpy(y).cuda()
    for m in CONT:
        got = pl.predictive_metric_from_matrix(tx, tl, y, m, 0.9)
        np.testing.assert_allclose(pair(got), pair(orc.loo_predictive_metric_arrays(x, ll, y, m, 0.9)), rtol=1e-9)
    for scale in (False, True):
        np.random.seed(5)
        pw, k = pl.score_from_matrix(tx, tx, ty, tl, 0.9, permutations=1, scale=scale)
        np.random.seed(5)
        want, wk = orc.loo_score_arrays(x, x, y, ll, 0.9, 1, scale)
        np.testing.assert_allclose(pw, want, rtol=1e-9)
        np.testing.assert_allclose(k, wk, rtol=1e-9, atol=1e-10)
