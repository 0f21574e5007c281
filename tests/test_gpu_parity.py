"""GPU parity tests proper: the HIP engine (through the C ABI) against the golden vectors of
the real reference and against the CPU oracle on seeded inputs.  Run with ``-m gpu``.

Tolerance: north_star asks for 1e-6 relative on fp64 inputs; the engine is held to RTOL
below (three orders tighter) with an absolute floor ATOL for quantities that pass through 0."""

import os
import warnings

import numpy as np
import pytest

import cases
from conftest import load_golden
from oracle import psis_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RTOL = 1e-9
ATOL = 1e-10


@pytest.fixture(scope="module")
def eng():
    from pyloo_amd.engine import get_engine

    return get_engine(0)


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    # identical NaN / inf patterns are part of the in-band contract
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    inf = np.isinf(b)
    assert np.array_equal(np.isinf(a), inf) and np.array_equal(a[inf], b[inf]), f"{what}: inf pattern differs"
    ok = np.isfinite(b)
    np.testing.assert_allclose(a[ok], b[ok], rtol=rtol, atol=atol, err_msg=what)


def has_tail_ties(ll_row, M):
    """True when two draws of the PSIS tail share the same log ratio."""
    x = -ll_row.astype(np.float64)
    if not np.all(np.isfinite(x)):
        return False
    x = x - x.max()
    cut = max(np.sort(x)[-M - 1], orc.LOG_TINY)
    tail = x[x > cut]
    return tail.size != np.unique(tail).size


@pytest.mark.parametrize("case", [c[0] for c in cases.CASES])
def test_golden_loo_pass(eng, case):
    g = load_golden(case)
    ll, reff = g["ll"], float(g["reff"])
    M = orc.tail_count(ll.shape[1], reff)
    res = eng.psis_loo(ll, M, "psis", 1.0, float(g["good_k"]))
    close(res["diag"], g["khat"], what="khat")
    close(res["loo_i"], g["loo_i"], what="loo_i")
    close(res["lppd_i"], g["lppd_i"], what="lppd_i")


@pytest.mark.parametrize("case", [c[0] for c in cases.CASES])
def test_golden_weights(eng, case):
    g = load_golden(case)
    ll, reff = g["ll"], float(g["reff"])
    M = orc.tail_count(ll.shape[1], reff)
    lw, k = eng.importance_weights(-ll, M, "psis")
    assert lw.dtype == ll.dtype and k.dtype == np.float64
    close(k, g["khat"], what="khat")
    want = g["lw"]
    tied = [i for i in range(ll.shape[0]) if has_tail_ties(ll[i], M)]
    if tied:
        # Equal log ratios inside the tail: the reference hands the GPD quantiles to tied draws in
        # the order of NumPy's unstable argsort (psis.py:146); the engine uses draw order.  The
        # multiset of weights is identical, only which tied draw gets which quantile can differ.
        lw, want = lw.copy(), want.copy()
        for i in tied:
            lw[i], want[i] = np.sort(lw[i]), np.sort(want[i])
    if ll.dtype == np.float32:
        # parity target for f32 input is the reference on the f64-upcast data, rounded to f32
        close(lw, want.astype(np.float32), rtol=2e-7, atol=1e-7, what="lw")
    else:
        close(lw, want, what="lw")
    ok = ~np.isnan(lw).any(axis=1)
    np.testing.assert_allclose(np.exp(lw[ok].astype(np.float64)).sum(axis=1), 1.0, rtol=1e-5 if ll.dtype == np.float32 else 1e-12)


def test_known_answer(eng):
    g = load_golden("known_answer_s4000")
    res = eng.psis_loo(g["ll"], 190, "psis", 1.0, 0.7)
    close(res["diag"], [0.1263015392349325, 0.31231166382304587, 0.4983134530821454,
                        0.6843220218846074, 0.8703208341622388, 1.1492715840633747], what="khat")
    agg = res["agg"]
    np.testing.assert_allclose(agg[1], -7.495953310386053, rtol=RTOL)
    np.testing.assert_allclose(np.sqrt(agg[2]), 2.6906896870466506, rtol=RTOL)
    np.testing.assert_allclose(agg[3] - agg[1], 4.771879530745729, rtol=RTOL)
    np.testing.assert_allclose(np.sqrt(agg[2] / 6), 1.0984694649055418, rtol=RTOL)
    assert agg[0] == 6 and agg[4] == 2


@pytest.mark.parametrize("S,N,reff,dt", [(4000, 96, 1.0, np.float64), (1000, 64, 0.6, np.float64),
                                         (257, 40, 1.0, np.float64), (64, 33, 2.0, np.float64),
                                         (6000, 24, 1.0, np.float64), (20000, 12, 1.0, np.float32),
                                         (4000, 50, 0.25, np.float32),
                                         # tail counts 230 / 222 / 124: the four- and two-quad variants of the fit kernel
                                         (4096, 40, 0.7, np.float64), (3000, 30, 0.55, np.float32), (2048, 37, 1.2, np.float64)])
def test_seeded_vs_oracle(eng, S, N, reff, dt):
    rng = np.random.default_rng(S * 7 + N)
    k = rng.uniform(0.05, 1.2, size=N)
    ll = (-k[:, None] * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))).astype(dt)
    ref = orc.loo_arrays(ll.astype(np.float64), reff)
    M = orc.tail_count(S, reff)
    res = eng.psis_loo(ll, M, "psis", 1.0, ref["good_k"])
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    agg = res["agg"]
    np.testing.assert_allclose(agg[1], ref["elpd_loo"], rtol=RTOL)
    np.testing.assert_allclose(np.sqrt(agg[2]), ref["se"], rtol=1e-8)
    np.testing.assert_allclose(agg[3], ref["lppd"], rtol=RTOL)
    assert int(agg[4]) == ref["n_high_k"] and int(agg[0]) == N
    lw, kk = eng.importance_weights(-ll, M, "psis")
    close(kk, ref["khat"], what="khat(lw)")
    if dt == np.float64:
        close(lw, ref["lw"], what="lw")


@pytest.mark.parametrize("scale_value", [1.0, -1.0, -2.0])
def test_scales(eng, scale_value):
    g = load_golden("s2000_r1_f64")
    ok = g["agg_rows"]
    ll = g["ll"][ok]
    res = eng.psis_loo(ll, 135, "psis", scale_value, float(g["good_k"]))
    want = orc.loo_aggregate(g["loo_i"][ok], g["lppd_i"][ok], g["khat"][ok], 2000, scale_value)
    close(res["loo_i"], scale_value * g["loo_i"][ok], what="loo_i")
    np.testing.assert_allclose(res["agg"][1], want["elpd_loo"], rtol=RTOL)
    np.testing.assert_allclose(np.sqrt(res["agg"][2]), want["se"], rtol=1e-8)
    np.testing.assert_allclose(res["agg"][3] - res["agg"][1] / scale_value, want["p_loo"], rtol=1e-8, atol=1e-6)


@pytest.mark.parametrize("method", ["sis", "tis"])
def test_sis_tis(eng, method):
    u = load_golden("units")
    v = u["lse_in"]
    lw, ess = eng.importance_weights(-v, 0, method)
    close(lw, u[f"{method}_lw"], what="lw")
    close(ess, u[f"{method}_ess"], what="ess")
    rng = np.random.default_rng(5)
    ll = -0.6 * rng.exponential(size=(37, 1500)) - 1.0
    ref = orc.loo_pointwise(ll, 1.0, method)
    res = eng.psis_loo(ll, 0, method, 1.0, 0.7)
    close(res["diag"], ref["diag"], what="ess")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    assert res["agg"][6] == pytest.approx(ref["diag"].min(), rel=1e-9)


@pytest.mark.parametrize("method", ["sis", "tis"])
@pytest.mark.parametrize("S,dt", [(4000, np.float64), (3998, np.float64), (1024, np.float32), (256, np.float64)])
def test_sis_tis_streaming_kernel(eng, method, S, dt):
    """SIS / TIS LOO pass on the streaming wave kernel (pla_is.h), rows for the general kernel mixed in;
    heavy-tailed rows so that the TIS truncation actually bites."""
    rng = np.random.default_rng(S)
    N = 150
    k = rng.uniform(0.05, 1.5, size=N)
    ll = -k[:, None] * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))
    ll[2, 7] = np.nan
    ll[3, S - 1] = -np.inf
    ll[4] *= 300.0                      # range above 690 nats
    ll[5] = -0.75                       # constant row: every weight 1/S
    ll[6, S // 3] = ll[6].min() - 60.0  # one dominant draw
    ll = ll.astype(dt)
    ref = orc.loo_pointwise(ll.astype(np.float64), 1.0, method)
    res = eng.psis_loo(ll, 0, method, 1.0, 0.7)
    close(res["diag"], ref["diag"], what="ess")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    assert 2 <= int(res["agg"][7]) <= 6          # nan, -inf (+ the wide row and the dominant draw when their range exceeds 690 nats)
    if method == "tis":
        trunc = (np.exp(ref["lw"]).max(axis=1) < np.exp(orc.importance_weights(-ll.astype(np.float64), "sis", 1.0)[0]).max(axis=1) * 0.999)
        assert trunc.sum() > 20                   # the truncation changed the largest weight in many rows


def test_device_tensors_match_host(eng):
    import torch

    rng = np.random.default_rng(11)
    ll = -0.5 * rng.exponential(size=(70, 4000)) - 2.0
    host = eng.psis_loo(ll, 190, "psis", 1.0, 0.7)
    t = torch.from_numpy(ll).cuda()
    dev = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert np.array_equal(dev[key].cpu().numpy(), host[key]), key  # same kernels: bitwise equal
    lw_h, k_h = eng.importance_weights(-ll, 190)
    lw_d, k_d = eng.importance_weights(-t, 190)
    assert np.array_equal(lw_d.cpu().numpy(), lw_h) and np.array_equal(k_d.cpu().numpy(), k_h)
    # strided (non-contiguous rows) device input
    big = torch.zeros((70, 4100), dtype=torch.float64, device="cuda")
    big[:, :4000] = t
    view = big[:, :4000]
    dev2 = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
    close(dev2["loo_i"].cpu().numpy(), host["loo_i"], what="strided rows")
    close(dev2["diag"].cpu().numpy(), host["diag"], what="strided rows k")
    # obs-fastest layout (ArviZ native): handled through element strides, same numbers
    tt = t.t().contiguous().t()
    assert tt.stride(1) != 1
    dev3 = eng.psis_loo(tt, 190, "psis", 1.0, 0.7)  # general kernel (not S-contiguous)
    close(dev3["loo_i"].cpu().numpy(), host["loo_i"], what="obs-fastest")
    close(dev3["diag"].cpu().numpy(), host["diag"], what="obs-fastest k")


def test_input_not_modified_and_errors(eng):
    from pyloo_amd._capi import EngineError

    rng = np.random.default_rng(3)
    ll = rng.normal(size=(5, 300))
    keep = ll.copy()
    eng.psis_loo(ll, 52, "psis", 1.0, 0.7)
    eng.importance_weights(ll, 52)
    assert np.array_equal(ll, keep)
    with pytest.raises(EngineError):
        eng.psis_loo(ll, 300, "psis", 1.0, 0.7)  # M + 1 > S
    out = eng.psis_loo(np.zeros((0, 300)), 52, "psis", 1.0, 0.7)
    assert out["diag"].shape == (0,)


def test_fronts(eng):
    import pyloo_amd as pl

    g = load_golden("shapes")
    lw, k = pl.psislw(g["x1"], 0.7)
    assert isinstance(k, np.ndarray) and k.shape == () and lw.shape == g["x1"].shape
    close(lw, g["lw1"], what="lw1")
    close(k, g["k1"], what="k1")
    lw, k = pl.psislw(g["x3"], 0.7)
    assert lw.shape == (2, 3, 100) and k.shape == (2, 3)
    close(lw, g["lw3"], what="lw3")
    close(k, g["k3"], what="k3")
    lw, k = pl.psislw(g["small"])
    assert k == np.inf
    close(lw, g["lw_small"], what="small")
    lw, k = pl.psislw(g["const"])
    assert k == np.inf
    close(lw, g["lw_const"], what="const")
    lw2, k2 = pl.compute_importance_weights(g["x3"], "psis", 0.7)
    assert np.array_equal(lw2, pl.psislw(g["x3"], 0.7)[0])
    with pytest.raises(ValueError, match="Invalid method 'nope'"):
        pl.compute_importance_weights(g["x3"], "nope")
    with pytest.raises(IndexError):
        pl.psislw(np.zeros(1))


def test_loo_front_matches_reference_numbers(eng):
    import pyloo_amd as pl

    g = load_golden("known_answer_s4000")
    ll = g["ll"]  # (6, 4000) -> (chain=4, draw=1000, obs=6)
    arr = np.moveaxis(ll.reshape(6, 4, 1000), 0, -1)
    with pytest.warns(UserWarning, match="Estimated shape parameter of Pareto distribution is greater than 0.70 for 2 observations"):
        res = pl.loo({"log_likelihood": {"obs": arr}}, pointwise=True, reff=1.0)
    assert list(res.index) == ["elpd_loo", "se", "p_loo", "p_loo_se", "n_samples", "n_data_points", "warning",
                               "loo_i", "scale", "looic", "looic_se", "pareto_k", "good_k", "subsample_size"]
    np.testing.assert_allclose(res["elpd_loo"], -7.495953310386053, rtol=RTOL)
    np.testing.assert_allclose(res["se"], 2.6906896870466506, rtol=RTOL)
    np.testing.assert_allclose(res["p_loo"], 4.771879530745729, rtol=RTOL)
    np.testing.assert_allclose(res["p_loo_se"], 1.0984694649055418, rtol=RTOL)
    np.testing.assert_allclose(res["looic"], 2 * 7.495953310386053, rtol=RTOL)
    assert res["n_samples"] == 4000 and res["n_data_points"] == 6 and res["warning"] and res["scale"] == "log"
    close(np.asarray(res["pareto_k"]), g["khat"], what="pareto_k")
    close(np.asarray(res["loo_i"]), g["loo_i"], what="loo_i")
    assert "Pareto k diagnostic values" in str(res)


def test_fast_and_slow_rows_in_one_matrix(eng):
    """Rows the wave kernel must hand to the general kernel (non-finite entries, > 690 nats of range,
    degenerate thresholds) mixed into ordinary rows: every row equals the oracle, and the hand-over
    count is what it should be."""
    rng = np.random.default_rng(2024)
    N, S = 300, 4000
    k = rng.uniform(0.05, 1.3, size=N)
    ll = -k[:, None] * rng.exponential(size=(N, S)) - 1.0
    special = {
        3: "nan", 17: "pinf", 40: "ninf", 77: "wide", 101: "const", 150: "two_values", 222: "outlier", 260: "grid",
    }
    ll[3, 5] = np.nan
    ll[17, 9] = np.inf
    ll[40, 11] = -np.inf
    ll[77] *= 400.0                      # range far above 690 nats: the log(DBL_MIN) floor binds
    ll[101] = -2.5                       # constant row: empty tail, k = inf
    ll[150] = np.where(np.arange(S) % 2 == 0, -1.0, -3.0)   # two distinct values: ties everywhere
    ll[222, 7] = -5000.0                 # one draw 5000 nats below the rest
    ll[260] = -np.round(ll[260] * -4.0) / 4.0                # coarse grid: ties inside the tail
    ref = orc.loo_arrays(ll, 1.0)
    res = eng.psis_loo(ll, 190, "psis", 1.0, 0.7)
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    n_slow = int(res["agg"][7])
    assert 5 <= n_slow <= 16, n_slow     # nan, pinf, ninf, wide, const, outlier (+ possibly tie rows)
    assert int(res["agg"][5]) == int(np.sum(~np.isfinite(ref["khat"])))


@pytest.mark.parametrize("S", [4000, 8000])
def test_rows_with_a_dominant_draw(eng, S):
    """One draw far above the rest: the smoothed tail is much lighter than the raw one, the regime where
    'sum of all exponentials minus the tail's' cancels.  Those rows must still equal the reference."""
    rng = np.random.default_rng(3)
    mags = [0.0, 2.0, 5.0, 8.0, 10.0, 15.0, 20.0, 40.0, 100.0, 300.0, 600.0]
    ll = -0.5 * rng.exponential(size=(len(mags), S)) - 1.0
    for i, g in enumerate(mags):
        if g:
            ll[i, (S // 2 + 997 * i) % S] = ll[i].min() - g
    M = orc.tail_count(S, 1.0)
    ref = orc.loo_arrays(ll, 1.0)
    res = eng.psis_loo(ll, M, "psis", 1.0, 0.7)
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    lw_ref, _ = orc.psislw(-ll, 1.0)
    lw, _ = eng.importance_weights(-ll, M, "psis")
    close(lw, lw_ref, what="lw")


def test_weights_fast_and_slow_rows_in_one_matrix(eng):
    """psislw through the wave kernel (weights mode) with rows it has to hand to the general kernel mixed
    in: every row's smoothed log-weights and k-hat equal the oracle; f32 keeps its dtype."""
    rng = np.random.default_rng(515)
    N, S = 200, 4000
    k = rng.uniform(0.05, 1.3, size=N)
    logw = k[:, None] * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))
    logw[5, 3] = -np.inf                  # weight 0 for one draw
    logw[9, 100] = np.nan                 # NaN row: k = inf, NaN weights
    logw[31] = 1.25                       # constant row
    logw[64] *= 500.0                     # range above 690 nats
    logw[120] = np.round(logw[120] * 8.0) / 8.0   # coarse grid: ties inside the tail
    lw_ref, k_ref = orc.psislw(logw, 1.0)
    lw, kk = eng.importance_weights(logw, 190, "psis")
    close(kk, k_ref, what="khat")
    tied = [i for i in range(N) if has_tail_ties(-logw[i], 190)]
    a, b = lw.copy(), lw_ref.copy()
    for i in tied:
        a[i], b[i] = np.sort(a[i]), np.sort(b[i])
    close(a, b, what="lw")
    ok = np.isfinite(k_ref)
    np.testing.assert_allclose(np.exp(lw[ok]).sum(axis=1), 1.0, rtol=1e-12)
    # S not a multiple of 64 lanes x 16 bytes: partial last vector
    logw2 = logw[:40, :3998].copy()
    lw2_ref, k2_ref = orc.psislw(logw2, 1.0)
    lw2, k2 = eng.importance_weights(logw2, orc.tail_count(3998, 1.0), "psis")
    close(k2, k2_ref, what="khat(3998)")
    t2 = [i for i in range(40) if has_tail_ties(-logw2[i], orc.tail_count(3998, 1.0))]
    for i in t2:
        lw2[i], lw2_ref[i] = np.sort(lw2[i]), np.sort(lw2_ref[i])
    close(lw2, lw2_ref, what="lw(3998)")
    # f32 in, f32 out
    l32 = logw[:64].astype(np.float32)
    ok32 = np.isfinite(l32).all(axis=1)
    lw32, k32 = eng.importance_weights(l32, 190, "psis")
    assert lw32.dtype == np.float32
    ref32, kref32 = orc.psislw(l32.astype(np.float64), 1.0)
    close(k32, kref32, what="khat(f32)")
    t32 = [i for i in range(64) if has_tail_ties(-l32[i].astype(np.float64), 190)]
    a32, b32 = lw32.copy(), ref32.astype(np.float32)
    for i in t32:
        a32[i], b32[i] = np.sort(a32[i]), np.sort(b32[i])
    close(a32, b32, rtol=2e-7, atol=1e-6, what="lw(f32)")


def test_weights_device_tensors_full_size(eng):
    """Weights mode on a device-resident matrix (S=4000 x N=20k): normalised rows, k-hat equal to the LOO
    pass on the negated matrix, bitwise reproducible."""
    import torch

    S, N = 4000, 20_000
    ll = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0002)
    logw = -ll
    lw, k = eng.importance_weights(logw, 190, "psis")
    lw_b, k_b = eng.importance_weights(logw, 190, "psis")
    res = eng.psis_loo(ll, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    assert torch.equal(lw, lw_b) and torch.equal(k, k_b)
    np.testing.assert_allclose(k.cpu().numpy(), res["diag"].cpu().numpy(), rtol=1e-9)  # two code paths (fused weights pass, split LOO pass)
    sums = torch.exp(lw).sum(dim=1).cpu().numpy()
    np.testing.assert_allclose(sums, 1.0, rtol=1e-12)
    # loo_i recomputed from the returned weights (loo.py:289,319-324) equals the fused pass
    loo_from_w = torch.logsumexp(lw + ll, dim=1).cpu().numpy()
    np.testing.assert_allclose(loo_from_w, res["loo_i"].cpu().numpy(), rtol=1e-9, atol=1e-10)
    idx = np.arange(0, N, 499)
    ref_lw, ref_k = orc.psislw(logw[idx].cpu().numpy(), 1.0)
    close(k.cpu().numpy()[idx], ref_k, what="khat")
    close(lw.cpu().numpy()[idx], ref_lw, what="lw")


def test_long_row_weights_take_the_split_pass(eng, monkeypatch):
    """psislw on device-resident rows longer than the registers (S > 4096, tails the fit kernel takes): selection kernel ->
    fit kernel -> output kernel (csrc/pla_lwout.h), in blocks of 2^17 observations.  Rows on both sides of the block boundary
    equal a separate call over just those rows bit for bit, sampled rows equal the oracle (tied tail draws as multisets),
    every row is normalised, k-hat equals the LOO pass's, and the call is reproducible."""
    import torch

    S, N = 4352, (1 << 17) + 37
    M = orc.tail_count(S, 1.0)
    ll = torch.empty((N, S), dtype=torch.float32, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0021, k_lo=0.05, k_hi=1.2)
    ll[5, : S // 2] = torch.round(ll[5, : S // 2] * 8.0) / 8.0          # ties, also inside the tail
    ll[(1 << 17) + 3, 100:2000] = torch.round(ll[(1 << 17) + 3, 100:2000] * 4.0) / 4.0
    ll[9, 17] = float("nan")                                            # rows for the general kernel
    ll[(1 << 17) - 1] *= 300.0                                         # (a range above 690 nats)
    logw = -ll
    lw, k = eng.importance_weights(logw, M, "psis")
    assert "lw_output_kernel" in eng.last_kernels(), eng.last_kernels()
    lw_b, k_b = eng.importance_weights(logw, M, "psis")
    torch.cuda.synchronize()

    def same_rows(a, b):
        # equal tail draws may swap their quantiles between two runs (a counter hands out their ranks: f32 rows tie now and
        # then): such rows must hold the same multiset; nothing else may differ
        same = (a == b) | (torch.isnan(a) & torch.isnan(b))
        bad = (~same).any(dim=1).nonzero().flatten()
        assert bad.numel() <= 0.05 * a.shape[0], bad.numel()
        if bad.numel():
            sa, sb = torch.sort(a[bad], dim=1).values, torch.sort(b[bad], dim=1).values
            assert torch.equal(sa.nan_to_num(9.0), sb.nan_to_num(9.0))

    same_rows(lw, lw_b)
    assert torch.equal(k.nan_to_num(7.0), k_b.nan_to_num(7.0))
    lo, hi = (1 << 17) - 40, (1 << 17) + 37
    part, k_part = eng.importance_weights(logw[lo:hi].clone(), M, "psis")
    same_rows(lw[lo:hi], part)
    assert torch.equal(k[lo:hi].nan_to_num(7.0), k_part.nan_to_num(7.0))
    # EVERY row against the general kernel (all 26 M patched positions: the output kernel writes a draw's smoothed weight right
    # behind the row's own store of that position, without waiting for it -- a patch that lost that race would show here)
    monkeypatch.setenv("PLA_FORCE_PATH", "1")
    lw_g, k_g = eng.importance_weights(logw, M, "psis")
    monkeypatch.delenv("PLA_FORCE_PATH")
    assert "lw_output_kernel" not in eng.last_kernels()
    near = torch.isclose(lw, lw_g, rtol=3e-7, atol=2e-7, equal_nan=True)
    off = (~near).any(dim=1).nonzero().flatten()
    assert off.numel() <= 0.05 * N, off.numel()
    if off.numel():  # (tied tail draws: the same multiset of weights)
        sa, sb = torch.sort(lw[off], dim=1).values, torch.sort(lw_g[off], dim=1).values
        assert bool(torch.isclose(sa, sb, rtol=3e-7, atol=2e-7, equal_nan=True).all())
    del lw_g
    res = eng.psis_loo(ll, M, "psis", 1.0, 0.7)
    fin = torch.isfinite(k)
    np.testing.assert_allclose(k[fin].cpu().numpy(), res["diag"][fin].cpu().numpy(), rtol=1e-9)
    good = ~torch.isnan(lw).any(dim=1)
    assert int((~good).sum()) == 1                                      # (the NaN row)
    sums = torch.exp(lw[good].double()).sum(dim=1).cpu().numpy()
    np.testing.assert_allclose(sums, 1.0, rtol=2e-5)                    # (f32 outputs)
    idx = np.unique(np.concatenate([np.arange(0, N, 2999), [5, 9, (1 << 17) - 1, 1 << 17, (1 << 17) + 3, N - 1]]))
    ref_lw, ref_k = orc.psislw(logw[idx].cpu().numpy().astype(np.float64), 1.0)
    got = lw.cpu().numpy()[idx].astype(np.float64)
    llh = ll.cpu().numpy()[idx]
    for i in range(len(idx)):
        if has_tail_ties(llh[i], M):
            got[i], ref_lw[i] = np.sort(got[i]), np.sort(ref_lw[i])
    close(k.cpu().numpy()[idx], ref_k, what="khat")
    close(got.astype(np.float32), ref_lw.astype(np.float32), rtol=3e-7, atol=2e-7, what="lw (f32 output)")


@pytest.mark.parametrize("S,N,reff,dt", [(8000, 150, 1.0, np.float64), (20000, 60, 1.0, np.float32),
                                         (4096 + 256, 40, 1.0, np.float64), (12288, 30, 0.5, np.float64),
                                         (4000, 80, 0.3, np.float64), (2000, 64, 0.35, np.float64)])
def test_chunked_kernel_vs_oracle(eng, S, N, reff, dt):
    """Rows longer than one register chunk and / or tail counts above 250 (pla_chunked.h), with rows the
    kernel must hand to the general one mixed in, and the row maximum placed in a late chunk."""
    rng = np.random.default_rng(S + N)
    k = rng.uniform(0.05, 1.3, size=N)
    ll = (-k[:, None] * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))).astype(np.float64)
    ll[1, S - 7] = ll[1].min() - 3.0          # the largest -ll of the row sits in the last chunk
    ll[2, S // 2] = ll[2].min() - 40.0        # ... far above the provisional shift
    ll[3, 5] = np.nan
    ll[4, S - 1] = -np.inf
    ll[5] *= 300.0                            # range above 690 nats
    ll[6] = -1.0                              # constant row
    ll[7, : S // 2] = np.round(ll[7, : S // 2] * 4.0) / 4.0   # ties
    ll = ll.astype(dt)
    ref = orc.loo_arrays(ll.astype(np.float64), reff)
    M = orc.tail_count(S, reff)
    res = eng.psis_loo(ll, M, "psis", 1.0, ref["good_k"])
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    n_slow = int(res["agg"][7])
    assert 4 <= n_slow <= 12, n_slow          # nan, -inf, wide, constant, dominant draw (+ threshold misses / ties)
    np.testing.assert_allclose(res["agg"][1], ref["elpd_loo"], rtol=RTOL)
    # weights mode of the same kernel (psislw on long rows): the row passes through the registers a second time
    lw, kk = eng.importance_weights(-ll, M, "psis")
    close(kk, ref["khat"], what="khat(lw)")
    want = ref["lw"].copy()
    for i in range(N):                        # tied tail draws: same multiset of weights, their order is the sort's (see above)
        if has_tail_ties(ll[i], M):
            lw[i], want[i] = np.sort(lw[i]), np.sort(want[i])
    if dt == np.float64:
        close(lw, want, what="lw")
    else:
        close(lw, want.astype(np.float32), rtol=2e-7, atol=1e-7, what="lw (f32 output)")
    ok = ~np.isnan(lw).any(axis=1)
    np.testing.assert_allclose(np.exp(lw[ok].astype(np.float64)).sum(axis=1), 1.0, rtol=1e-5 if dt == np.float32 else 1e-10)


def test_general_kernel_agrees_with_fast_path():
    """Same matrix through the wave kernel and (in a child process) through the general kernel only."""
    import subprocess
    import sys
    import tempfile

    rng = np.random.default_rng(77)
    ll = -rng.uniform(0.1, 1.1, size=(200, 1)) * rng.exponential(size=(200, 2048)) - 0.5
    from pyloo_amd.engine import get_engine

    fast = get_engine(0).psis_loo(ll, 136, "psis", 1.0, 0.7)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "ll.npy"), ll)
        code = (
            "import sys, numpy as np; sys.path.insert(0, %r); from pyloo_amd.engine import get_engine;"
            "ll = np.load(%r); r = get_engine(0).psis_loo(ll, 136, 'psis', 1.0, 0.7);"
            "np.savez(%r, diag=r['diag'], loo_i=r['loo_i'], lppd_i=r['lppd_i'], agg=r['agg'])"
        ) % (ROOT, os.path.join(d, "ll.npy"), os.path.join(d, "out.npz"))
        env = dict(os.environ, PLA_FORCE_PATH="1")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
        gen = np.load(os.path.join(d, "out.npz"))
        assert fast["agg"][7] == 0 and gen["agg"][7] == 0
        for key in ("diag", "loo_i", "lppd_i"):
            close(fast[key], gen[key], what=key)
        np.testing.assert_allclose(fast["agg"][1:4], gen["agg"][1:4], rtol=1e-10)


def test_full_size_properties(eng):
    """BASELINE config C2 size (S=4000 x N=100k, device-generated): determinism, reductions,
    oracle parity on a strided sample, and the smoothed weights' normalisation."""
    import torch

    S, N = 4000, 100_000
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0002)
    a = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    b = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert torch.equal(a[key], b[key]), key          # bitwise reproducible
    loo_i, lppd_i, diag = (a[k].cpu().numpy() for k in ("loo_i", "lppd_i", "diag"))
    agg = a["agg"].cpu().numpy()
    assert np.all(np.isfinite(loo_i)) and np.all(np.isfinite(lppd_i)) and np.all(np.isfinite(diag))
    assert np.all(loo_i <= lppd_i + 1e-9)                 # leave-one-out density never beats the in-sample one
    import math

    np.testing.assert_allclose(agg[1], math.fsum(loo_i), rtol=1e-12)
    np.testing.assert_allclose(agg[3], math.fsum(lppd_i), rtol=1e-12)
    np.testing.assert_allclose(agg[2] / N, np.var(loo_i), rtol=1e-9)
    assert agg[0] == N and agg[4] == np.sum(diag > 0.7)
    assert agg[7] <= 0.002 * N                            # rows where the speculative threshold missed
    again = eng.reduce_pointwise(a["diag"], a["loo_i"], a["lppd_i"], 0.7).cpu().numpy()
    assert np.array_equal(again[:7], agg[:7])
    idx = np.arange(0, N, 997)                            # 101 rows against the oracle
    ref = orc.loo_arrays(t[idx].cpu().numpy(), 1.0)
    close(diag[idx], ref["khat"], what="khat sample")
    close(loo_i[idx], ref["loo_i"], what="loo_i sample")
    close(lppd_i[idx], ref["lppd_i"], what="lppd_i sample")
    lw, kk = eng.importance_weights(-t[:2048], 190)
    torch.cuda.synchronize()
    np.testing.assert_allclose(torch.exp(lw).sum(dim=1).cpu().numpy(), 1.0, rtol=1e-12)
    close(kk.cpu().numpy(), diag[:2048], what="khat (weights pass)")


def test_device_path_is_graph_capturable(eng):
    """The PLA_DEVICE path does no allocation and no synchronisation once the engine workspace is sized:
    a whole LOO pass (memsets, row kernels, reductions) is captured in a HIP graph and replayed on new data."""
    import torch

    S, N = 4000, 3000
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=11)
    warm = eng.psis_loo(t, 190, "psis", 1.0, 0.7)          # sizes the workspace, uploads the quantile tables
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    eng.fill_synthetic(t, seed=12)                          # new matrix in the same buffer
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    fresh = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert torch.equal(out[key], fresh[key]), key
    assert not torch.equal(out["loo_i"], warm["loo_i"])
    # frozen workspace (include/pyloo_amd.h, "HIP graphs"): a call that would have to reallocate fails instead of pulling the
    # buffers from under a captured graph.  On an engine of its own: the shared one has been sized by earlier tests.
    from pyloo_amd._capi import EngineError
    from pyloo_amd.engine import Engine

    own = Engine(0)
    try:
        small = own.psis_loo(t, 190, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
        own.set_frozen(True)
        bigger = torch.empty((4 * N, S), dtype=torch.float64, device="cuda")
        eng.fill_synthetic(bigger, seed=13)
        with pytest.raises(EngineError) as err:
            own.psis_loo(bigger, 190, "psis", 1.0, 0.7)
        assert err.value.code == -6
        with pytest.raises(EngineError):
            own.psis_loo(t, 150, "psis", 1.0, 0.7)            # a tail count without a table yet: refused as well
        again = own.psis_loo(t, 190, "psis", 1.0, 0.7)        # what fits still runs
        torch.cuda.synchronize()
        assert torch.equal(again["loo_i"], small["loo_i"]) and torch.equal(again["loo_i"], fresh["loo_i"])
        own.set_frozen(False)
        own.psis_loo(bigger, 190, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
    finally:
        own.close()
    other = eng.psis_loo(t, 150, "psis", 1.0, 0.7)          # eager, another M: a second table
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["loo_i"], fresh["loo_i"]) and not torch.equal(other["loo_i"], fresh["loo_i"])


def test_loo_i_front(eng):
    """loo_i (loo_i.py:16-294) through the real engine: the one row's weights come from the GPU."""
    import pyloo_amd as pl

    rng = np.random.default_rng(9)
    ll = -0.45 * rng.exponential(size=(6, 4000)) - 2.0
    d = {"log_likelihood": {"obs": np.moveaxis(ll.reshape(6, 4, 1000), 0, -1)}, "posterior": {"mu": np.zeros((4, 1000))}}
    res = pl.loo_i(4, d, reff=0.9, pointwise=True)
    lw, diag = orc.importance_weights(-ll[4][None, :], "psis", 0.9)
    lwll = lw + ll[4][None, :]
    want = orc.lse(lwll[0])
    np.testing.assert_allclose(res["elpd_loo"], want, rtol=RTOL)
    np.testing.assert_allclose(np.asarray(res["pareto_k"]).ravel(), diag, rtol=RTOL)
    w = np.exp(lwll - lwll.max())
    w /= w.sum()
    se = np.sqrt(np.log1p(np.sum(w**2 * (np.exp(ll[4]) - np.exp(want)) ** 2) / 0.9 / np.exp(want) ** 2))
    np.testing.assert_allclose(res["se"], se, rtol=1e-8)
    full = pl.loo(d, reff=0.9, pointwise=True)
    np.testing.assert_allclose(np.asarray(full["loo_i"]).ravel()[4], want, rtol=RTOL)


def test_observation_fastest_device_layout(eng):
    """(S, N) buffer viewed as (N, S) -- the ArviZ-native layout, read in place by the lane-per-observation kernels (LOO) or
    transposed block by block inside the library (weights)."""
    import torch

    rng = np.random.default_rng(21)
    ll = -0.5 * rng.exponential(size=(500, 4000)) - 1.0
    ref = orc.loo_arrays(ll, 1.0)
    buf = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda()   # (S, N), observations fastest
    view = buf.T
    assert view.stride(1) != 1
    res = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
    close(res["diag"].cpu().numpy(), ref["khat"], what="khat")
    close(res["loo_i"].cpu().numpy(), ref["loo_i"], what="loo_i")
    assert int(res["agg"][7].item()) == 0                       # the wave kernel took every row
    lw, k = eng.importance_weights(-view, 190, "psis")
    close(lw.cpu().numpy(), ref["lw"], what="lw")


@pytest.mark.parametrize("N,S,dt", [(500, 4000, np.float64), (130, 1000, np.float32), (67, 258, np.float64), (2, 4096, np.float64),
                                    (1000, 8000, np.float32)])
def test_observation_fastest_ingestion(eng, N, S, dt):
    """Observations-fastest device matrices are read in place with one lane per observation (LOO: pla_col.h where the shape
    allows, else the transposing ingestion, bitwise; WAIC: waic_col_kernel) and agree with the draws-fastest copy to rounding
    (another order of summation).  Ragged workgroups, both dtypes."""
    import torch

    rng = np.random.default_rng(N + S)
    ll = (-rng.uniform(0.1, 0.9, size=(N, 1)) * rng.exponential(size=(N, S)) - 0.5).astype(dt)
    M = orc.tail_count(S, 1.0)
    rowmajor = torch.from_numpy(ll).cuda()
    view = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T   # (N, S) view of an (S, N) buffer
    assert view.stride(0) == 1 and view.stride(1) == N
    a, b = eng.psis_loo(view, M, "psis", 1.0, 0.7), eng.psis_loo(rowmajor, M, "psis", 1.0, 0.7)
    column_path = S >= 512 and M <= 250
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        if column_path:
            n_cmp = 7 if key == "agg" else None
            np.testing.assert_allclose(a[key].cpu().numpy()[:n_cmp], b[key].cpu().numpy()[:n_cmp], rtol=1e-11, atol=1e-12, err_msg=key)
        else:
            np.testing.assert_array_equal(a[key].cpu().numpy(), b[key].cpu().numpy(), err_msg=key)
    assert a["agg"][7].item() == 0
    wa, wb = eng.waic(view, 1.0), eng.waic(rowmajor, 1.0)  # (one lane per observation as well: pla_waic.h, waic_col_kernel)
    for key in ("lppd_i", "var_i", "waic_i", "agg"):
        np.testing.assert_allclose(wa[key].cpu().numpy(), wb[key].cpu().numpy(), rtol=1e-11, atol=1e-12, err_msg=key)
    sub = eng.psis_loo(view, M, "psis", 1.0, 0.7, rows=np.array([N - 1, 0]))   # (row selection on such a view: copied first)
    np.testing.assert_array_equal(sub["loo_i"].cpu().numpy(), b["loo_i"].cpu().numpy()[[N - 1, 0]])


@pytest.mark.parametrize("N,S,dt", [(300, 4000, np.float64), (65, 1000, np.float32), (2, 256, np.float64)])
def test_host_observation_fastest_view(eng, N, S, dt):
    """A (chain, draw, obs) host array viewed as (obs, sample) -- what stack_samples() hands on -- goes up as pitched slabs
    and is transposed on the device: same bits as the host-transposed copy, no copy made on the host."""
    rng = np.random.default_rng(N * S)
    native = (-rng.uniform(0.1, 0.9, size=(1, 1, N)) * rng.exponential(size=(4, S // 4, N)) - 0.5).astype(dt)  # (chain, draw, obs)
    view = native.reshape(S, N).T
    assert view.base is not None and view.strides == (native.itemsize, N * native.itemsize)
    copy = np.ascontiguousarray(view)
    M = orc.tail_count(S, 1.0)
    a, b = eng.psis_loo(view, M, "psis", 1.0, 0.7), eng.psis_loo(copy, M, "psis", 1.0, 0.7)
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    wa, wb = eng.waic(view, 1.0), eng.waic(copy, 1.0)
    for key in ("lppd_i", "var_i", "waic_i", "agg"):
        np.testing.assert_array_equal(wa[key], wb[key], err_msg=key)
    import pyloo_amd as pl
    from pyloo_amd.utils import stack_samples

    m, _, _, _ = stack_samples(native)
    assert m.strides == view.strides and np.shares_memory(m, native)       # the front hands the view on
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = pl.loo(native, reff=1.0, pointwise=True)
    np.testing.assert_array_equal(np.asarray(out["loo_i"]).ravel(), b["loo_i"])


@pytest.mark.parametrize("N", [1, 2, 3, 5, 63, 65])
@pytest.mark.parametrize("S,dt", [(256, np.float64), (258, np.float64), (1000, np.float32), (4094, np.float64),
                                  (4096, np.float64), (4096, np.float32), (4100, np.float64), (8192, np.float32)])
def test_odd_shapes(eng, N, S, dt):
    """Row counts that do not fill a workgroup and draw counts at the edges of the register chunk."""
    rng = np.random.default_rng(N * 10007 + S)
    ll = (-rng.uniform(0.1, 0.9, size=(N, 1)) * rng.exponential(size=(N, S)) - 0.5).astype(dt)
    ref = orc.loo_arrays(ll.astype(np.float64), 1.0)
    M = orc.tail_count(S, 1.0)
    res = eng.psis_loo(ll, M, "psis", 1.0, ref["good_k"])
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    np.testing.assert_allclose(res["agg"][1], ref["elpd_loo"], rtol=RTOL)
    w = eng.waic(ll, 1.0)
    wr = orc.waic_arrays(ll.astype(np.float64), 1)
    np.testing.assert_allclose(w["waic_i"], wr["waic_i"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("S,reff", [(4000, 1.0), (2048, 1.2), (4096, 0.7)])
def test_fit_kernel_branches(eng, S, reff):
    """Rows that steer the fit kernel of the split pass off its straight-line path: ties at the cutoff that shorten the
    tail a little (p_j from the observation's own n, same grid) or a lot (another m_est: handed to the general kernel),
    repeated draws inside the tail, and tails of very different weight side by side in one wavefront."""
    rng = np.random.default_rng(S)
    N = 48
    M = orc.tail_count(S, reff)
    k = np.tile([0.05, 0.3, 0.7, 1.1], N // 4)[:, None]
    ll = -k * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))
    order = np.argsort(ll, axis=1)  # ascending ll = descending log ratio: the tail is the first M entries
    for i in range(N):
        o = order[i]
        if i % 4 == 1:      # the cutoff value repeated 3 times inside the tail: n = M - 3
            ll[i, o[M - 3:M]] = ll[i, o[M]]
        elif i % 4 == 2:    # ... 40 times: n = M - 40, isqrt changes for the tail counts used here
            ll[i, o[M - 40:M]] = ll[i, o[M]]
        elif i % 4 == 3:    # repeated draws inside the tail (a Metropolis chain that stood still)
            ll[i, o[5:9]] = ll[i, o[4]]
            ll[i, o[60:62]] = ll[i, o[59]]
    ref = orc.loo_arrays(ll, reff)
    res = eng.psis_loo(ll, M, "psis", 1.0, ref["good_k"])
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    np.testing.assert_allclose(res["agg"][1], ref["elpd_loo"], rtol=RTOL)


def test_heavy_tails_near_the_cancellation_guard(eng):
    """k-hat above 1: the raw tail carries most of the row and total = (sum_all - sum_tail) + sum_smoothed cancels by up to
    the guard (PLA_CANCEL_GUARD) before a row is handed to the general kernel; rows on both sides of it stay within tolerance."""
    rng = np.random.default_rng(77)
    N, S = 768, 4000
    k = rng.uniform(1.0, 1.6, size=(N, 1))
    ll = -k * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))
    ref = orc.loo_arrays(ll, 1.0)
    res = eng.psis_loo(ll, orc.tail_count(S, 1.0), "psis", 1.0, ref["good_k"])
    close(res["diag"], ref["khat"], what="khat")
    close(res["loo_i"], ref["loo_i"], what="loo_i")
    close(res["lppd_i"], ref["lppd_i"], what="lppd_i")
    assert 0 < int(res["agg"][7]) < N // 4  # some rows are beyond the guard, most are not


def test_ingestion_in_many_blocks():
    """The transposing ingestion (PLA_INGEST_TRANSPOSE=1: the LOO pass too) with 1 MB blocks (32 rows each at S = 4000): block
    boundaries, output offsets and the hand-over workspace of the split pass sized per block.  Own process: both knobs are
    read once per process."""
    import subprocess
    import sys

    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from pyloo_amd.engine import get_engine
from oracle import psis_oracle as orc
eng = get_engine(0)
rng = np.random.default_rng(5)
N, S = 301, 4000
ll = -rng.uniform(0.1, 0.9, size=(N, 1)) * rng.exponential(size=(N, S)) - 0.5
view = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T
a = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
b = eng.psis_loo(torch.from_numpy(ll).cuda(), 190, "psis", 1.0, 0.7)
for k in ("diag", "loo_i", "lppd_i", "agg"):
    assert np.array_equal(a[k].cpu().numpy(), b[k].cpu().numpy()), k
w1, w2 = eng.waic(view, 1.0), eng.waic(torch.from_numpy(ll).cuda(), 1.0)
assert np.array_equal(w1["waic_i"].cpu().numpy(), w2["waic_i"].cpu().numpy())
lw1, k1 = eng.importance_weights(-view, 190, "psis")
lw2, k2 = eng.importance_weights(torch.from_numpy(-ll).cuda(), 190, "psis")
assert np.array_equal(lw1.cpu().numpy(), lw2.cpu().numpy()) and np.array_equal(k1.cpu().numpy(), k2.cpu().numpy())
h1 = eng.psis_loo(np.ascontiguousarray(ll.T).T, 190, "psis", 1.0, 0.7)   # host view, observations fastest
assert np.array_equal(h1["loo_i"], b["loo_i"].cpu().numpy())
print("blocks ok")
""" % ROOT
    env = dict(os.environ, PLA_INGEST_BLOCK_MB="1", PLA_INGEST_TRANSPOSE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "blocks ok" in out.stdout, out.stdout + out.stderr


def test_column_kernels_on_observation_fastest_matrices(eng):
    """pla_col.h: one lane per observation on an (S, N) buffer -- several blocks of observations (> 262 144), a ragged last
    workgroup, heavy-tailed rows, rows with non-finite draws (general kernel through the strided view), against the
    draws-fastest pass and the oracle."""
    import torch

    N, S = 300_011, 512
    t = torch.empty((N, S), dtype=torch.float32, device="cuda")
    eng.fill_synthetic(t, seed=99, k_lo=0.05, k_hi=1.1)
    t[7, 100] = float("nan")
    t[70_000, 3] = float("inf")
    t[262_200] = -3.0  # a constant row: every weight equal, k = inf (psis.py:142-144 through ties at the cutoff)
    M = orc.tail_count(S, 1.0)
    view = t.T.contiguous().T  # (N, S) view of an (S, N) buffer: observations fastest
    assert view.stride(0) == 1
    a = eng.psis_loo(view, M, "psis", 1.0, 0.7)
    b = eng.psis_loo(t, M, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i"):
        x, y = a[key].cpu().numpy(), b[key].cpu().numpy()
        assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y)), key
        ok = np.isfinite(y)
        np.testing.assert_allclose(x[ok], y[ok], rtol=1e-10, atol=1e-11, err_msg=key)
    assert a["agg"][7].item() <= 0.01 * N
    idx = np.r_[0:40, 7, 70_000, 262_140:262_210, N - 30:N]
    ref = orc.loo_arrays(t[torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.float64), 1.0)
    close(a["diag"].cpu().numpy()[idx], ref["khat"], what="khat")
    close(a["loo_i"].cpu().numpy()[idx], ref["loo_i"], what="loo_i")
    close(a["lppd_i"].cpu().numpy()[idx], ref["lppd_i"], what="lppd_i")


@pytest.mark.gpu
@pytest.mark.parametrize("N,S,reff", [(300_011, 4000, 1.0), (60_007, 4000, 0.7), (100_003, 1000, 1.0), (50_001, 777, 1.0), (20_000, 4096, 0.7)])
def test_tile_kernel_on_observation_fastest_f64_matrices(eng, N, S, reff, monkeypatch):
    """pla_tile.h: a workgroup per 16 observations on an f64 (S, N) buffer -- a ragged last group, row lengths with steps behind
    the last whole batch and draws behind the last whole step, heavy-tailed rows, rows with non-finite draws and a constant row
    (general kernel through the strided view), against the draws-fastest pass and the oracle."""
    import torch

    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=1234 + S, k_lo=0.05, k_hi=1.1)
    t[7, 100] = float("nan")
    t[N // 3, 3] = float("inf")
    t[N // 2] = -3.0  # a constant row: every weight equal, k = inf (psis.py:142-144 through ties at the cutoff)
    t[N - 1, S - 1] = -40.0  # the very last element of the matrix: by far the largest raw value of its row
    M = orc.tail_count(S, reff)
    view = t.T.contiguous().T  # (N, S) view of an (S, N) buffer: observations fastest
    assert view.stride(0) == 1
    a = eng.psis_loo(view, M, "psis", 1.0, 0.7)
    assert "tile_loo_kernel" in eng.last_kernels()
    streamed = "fit_rows_stream_kernel" in eng.last_kernels()
    # (the streamed pass has the shorter lists: tail counts of 227 and 230 at S ~ 4000 leave too little room around the expected count)
    assert streamed == (reff == 1.0), eng.last_kernels()
    monkeypatch.setenv("PLA_PIPE", "0")  # the two kernels back to back (longer lists, another threshold: the same tails)
    a0 = eng.psis_loo(view, M, "psis", 1.0, 0.7)
    assert "tile_loo_kernel" in eng.last_kernels() and "fit_rows_stream_kernel" not in eng.last_kernels()
    monkeypatch.delenv("PLA_PIPE")
    b = eng.psis_loo(t, M, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for res in (a, a0):
        for key in ("diag", "loo_i", "lppd_i"):
            x, y = res[key].cpu().numpy(), b[key].cpu().numpy()
            assert np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isinf(x), np.isinf(y)), key
            ok = np.isfinite(y)
            np.testing.assert_allclose(x[ok], y[ok], rtol=1e-10, atol=1e-11, err_msg=key)
        assert res["agg"][7].item() <= 0.01 * N
    a2 = eng.psis_loo(view, M, "psis", 1.0, 0.7)  # (lists are appended to in whatever order the waves arrive: the results may not depend on it)
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert torch.equal(a[key], a2[key]) or np.array_equal(a[key].cpu().numpy(), a2[key].cpu().numpy(), equal_nan=True), key
    idx = np.r_[0:40, 7, N // 3, N // 2 - 20:N // 2 + 20, N - 30:N]
    ref = orc.loo_arrays(t[torch.from_numpy(idx).cuda()].cpu().numpy(), reff)
    close(a["diag"].cpu().numpy()[idx], ref["khat"], what="khat")
    close(a["loo_i"].cpu().numpy()[idx], ref["loo_i"], what="loo_i")
    close(a["lppd_i"].cpu().numpy()[idx], ref["lppd_i"], what="lppd_i")


@pytest.mark.gpu
@pytest.mark.parametrize("S", [512, 513, 543, 767, 768, 769, 800, 1535, 1536, 2047, 3999, 4001, 4096, 5000])
def test_tile_kernel_row_lengths_and_group_counts(eng, S):
    """Row lengths around every boundary of the tile kernel's sweep (32 draws per step, rings of 24 steps, the remainders behind
    both, rows shorter than one ring) times observation counts around its groups of 16 and its 256 workgroups, against the
    draws-fastest pass on the same numbers."""
    import torch

    M = orc.tail_count(S, 1.0)
    for N in (1, 15, 16, 17, 255, 4096 + 3, 16 * 256 * 2 + 31):
        t = torch.empty((N, S), dtype=torch.float64, device="cuda")
        eng.fill_synthetic(t, seed=77 + S + N, k_lo=0.05, k_hi=1.0)
        view = t.T.contiguous().T if N > 1 else torch.as_strided(t.clone(), (N, S), (1, N))
        assert view.stride(0) == 1 or N == 1
        a = eng.psis_loo(view, M, "psis", 1.0, 0.7)
        used = eng.last_kernels()
        b = eng.psis_loo(t, M, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
        if N > 1:
            assert "tile_loo_kernel" in used, (S, N, used)
        for key in ("diag", "loo_i", "lppd_i"):
            np.testing.assert_allclose(a[key].cpu().numpy(), b[key].cpu().numpy(), rtol=1e-10, atol=1e-11, err_msg=f"{key} S={S} N={N}")


@pytest.mark.gpu
def test_observations_fastest_pass_is_graph_capturable(eng):
    """The streamed tile pass (two internal streams forked from and joined to the caller's, flags zeroed by a kernel of the
    pass) captured in a HIP graph and replayed on new data in the same buffer."""
    import torch

    S, N = 4000, 5003
    buf = torch.empty((S, N), dtype=torch.float64, device="cuda")
    view = buf.T
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=21)
    buf.copy_(t.T)
    warm = eng.psis_loo(view, 190, "psis", 1.0, 0.7)  # sizes the workspace, sets the kernel's LDS attribute
    assert "tile_loo_kernel<SYNC>" in eng.last_kernels()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
    eng.fill_synthetic(t, seed=22)
    buf.copy_(t.T)  # new matrix in the same buffer
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    fresh = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert torch.equal(out[key], fresh[key]), key
    assert not torch.equal(out["loo_i"], warm["loo_i"])


@pytest.mark.gpu
def test_tile_kernel_on_a_window_of_a_wider_buffer(eng):
    """Observations fastest, but the matrix is a window of a wider buffer: the draws lie further apart than there are
    observations and the first observation is not on a 128-byte boundary (pieces then straddle cache lines: slower, not wrong)."""
    import torch

    S, N, pad = 4000, 20_003, 7
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=4242, k_lo=0.05, k_hi=1.0)
    wide = torch.full((S, N + pad), -1.0, dtype=torch.float64, device="cuda")
    wide[:, 3:3 + N] = t.T
    view = wide[:, 3:3 + N].T
    assert view.stride(0) == 1 and view.stride(1) == N + pad and view.data_ptr() % 128 != 0
    a = eng.psis_loo(view, 190, "psis", 1.0, 0.7)
    assert "tile_loo_kernel" in eng.last_kernels()
    b = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i"):
        np.testing.assert_allclose(a[key].cpu().numpy(), b[key].cpu().numpy(), rtol=1e-10, atol=1e-11, err_msg=key)
