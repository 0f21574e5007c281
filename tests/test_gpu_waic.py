"""GPU parity of the WAIC pass (pla_waic, through the C ABI) against the CPU oracle; run with ``-m gpu``."""

import numpy as np
import pytest

import cases
from conftest import load_golden
from oracle import psis_oracle as orc

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope="module")
def eng():
    from pyloo_amd.engine import get_engine

    return get_engine(0)


def check(res, want, n):
    np.testing.assert_allclose(res["lppd_i"], want["lppd_i"], rtol=RTOL, atol=1e-12)
    np.testing.assert_allclose(res["var_i"], want["var_i"], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(res["waic_i"], want["waic_i"], rtol=RTOL, atol=1e-10)
    agg = res["agg"]
    assert agg[0] == n
    np.testing.assert_allclose(agg[1], want["elpd_waic"], rtol=1e-10)
    np.testing.assert_allclose(np.sqrt(agg[2]), want["se"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(agg[3], want["p_waic"], rtol=1e-10)
    assert int(agg[4]) == int(np.sum(want["var_i"] > 0.4))


@pytest.mark.parametrize("case", [c[0] for c in cases.CASES])
def test_golden_inputs(eng, case):
    """The golden input matrices (edge rows included: NaN, +-inf, constant rows, 1e10 extremes)."""
    ll = load_golden(case)["ll"]
    want = orc.waic_arrays(ll.astype(np.float64), 1)
    res = eng.waic(ll, 1.0)
    check(res, want, ll.shape[0])
    assert int(res["agg"][7]) == int(np.sum(~np.isfinite(ll)))


@pytest.mark.parametrize("S,N,dt,scale", [(4000, 200, np.float64, 1.0), (4000, 64, np.float32, -2.0),
                                          (3998, 50, np.float64, -1.0), (1000, 40, np.float64, 1.0),
                                          (257, 30, np.float64, 1.0), (20000, 10, np.float32, 1.0),
                                          (64, 20, np.float64, 1.0), (7, 5, np.float64, 1.0)])
def test_seeded_vs_oracle(eng, S, N, dt, scale):
    rng = np.random.default_rng(S + N)
    ll = (-rng.uniform(0.05, 1.2, size=(N, 1)) * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))).astype(dt)
    want = orc.waic_arrays(ll.astype(np.float64), scale)
    check(eng.waic(ll, scale), want, N)


def test_strided_and_device(eng):
    import torch

    rng = np.random.default_rng(5)
    ll = -0.4 * rng.exponential(size=(300, 4000)) - 2.0
    want = orc.waic_arrays(ll, 1)
    t = torch.from_numpy(ll).cuda()
    res = {k: v.cpu().numpy() for k, v in eng.waic(t, 1.0).items()}
    check(res, want, 300)
    tt = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T  # draws-fastest view of an (S, N) buffer
    res = {k: v.cpu().numpy() for k, v in eng.waic(tt, 1.0).items()}
    check(res, want, 300)


def test_front_and_full_size(eng):
    import torch

    import pyloo_amd as pl

    S, N = 4000, 50_000
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0002)
    out = pl.waic_from_matrix(t, pointwise=True)
    again = pl.waic_from_matrix(t, pointwise=True)
    assert out["elpd_waic"] == again["elpd_waic"] and out["se"] == again["se"]        # reproducible
    wi = out["waic_i"].cpu().numpy()
    import math

    np.testing.assert_allclose(out["elpd_waic"], math.fsum(wi), rtol=1e-12)
    np.testing.assert_allclose(out["se"], np.sqrt(N * np.var(wi)), rtol=1e-9)
    idx = np.arange(0, N, 251)
    want = orc.waic_arrays(t[idx].cpu().numpy(), 1)
    np.testing.assert_allclose(wi[idx], want["waic_i"], rtol=RTOL, atol=1e-10)
    # lppd_i of the WAIC pass and of the LOO pass are the same quantity (waic.py:137-143 = loo.py:329-337)
    loo = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    w = eng.waic(t, 1.0)
    np.testing.assert_allclose(w["lppd_i"].cpu().numpy(), loo["lppd_i"].cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize("case", [c[0] for c in cases.CASES])
def test_golden_inputs_observation_fastest(eng, case):
    """The same golden matrices (NaN, +-inf, constant rows, 1e10 extremes) as (S, N) device buffers viewed as (N, S): the
    lane-per-observation WAIC kernel (waic_col_kernel) -- running maximum, batched variance, replacements counted."""
    import torch

    ll = load_golden(case)["ll"]
    want = orc.waic_arrays(ll.astype(np.float64), 1)
    view = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T
    assert view.stride(0) == 1 or ll.shape[0] == 1
    res = {k: v.cpu().numpy() for k, v in eng.waic(view, 1.0).items()}
    check(res, want, ll.shape[0])
    assert int(res["agg"][7]) == int(np.sum(~np.isfinite(ll)))


def test_observation_fastest_odd_shapes(eng):
    import torch

    rng = np.random.default_rng(77)
    for n, s, dt in ((257, 4001, np.float64), (1000, 13, np.float32), (3, 20000, np.float64)):
        ll = (-0.7 * rng.exponential(size=(n, s)) - 1.0).astype(dt)
        want = orc.waic_arrays(ll.astype(np.float64), -2)
        view = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T
        res = {k: v.cpu().numpy() for k, v in eng.waic(view, -2.0).items()}
        check(res, want, n)
