"""Row selection (``pla_psis_loo_rows`` / ``pla_waic_rows``) and the subsampled LOO front on the GPU:
against the same passes on a gathered copy, against the oracle, and against ``loo()`` when every observation is sampled."""

import warnings

import numpy as np
import pytest

from oracle import psis_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pyloo_amd.engine import get_engine

    return get_engine(0)


def make_ll(N, S, dt, seed):
    rng = np.random.default_rng(seed)
    k = rng.uniform(0.05, 0.9, size=(N, 1))
    return (-k * rng.exponential(size=(N, S)) + rng.normal(size=(N, 1))).astype(dt)


@pytest.mark.parametrize("S,dt,method", [(4000, np.float64, "psis"), (1000, np.float32, "psis"), (8000, np.float64, "psis"),
                                         (64, np.float64, "psis"), (4000, np.float64, "sis"), (2048, np.float32, "tis")])
@pytest.mark.parametrize("where", ["host", "device"])
def test_row_selection_equals_gathered_copy(eng, S, dt, method, where):
    import torch

    N = 57
    ll = make_ll(N, S, dt, S + N)
    rng = np.random.default_rng(1)
    idx = np.concatenate([rng.permutation(N)[:23], [5, 5, N - 1, 0]])  # unsorted, with repeats
    M = orc.tail_count(S, 1.0) if method == "psis" else 0
    src = torch.as_tensor(ll).cuda() if where == "device" else ll
    sub = torch.as_tensor(ll[idx]).cuda() if where == "device" else np.ascontiguousarray(ll[idx])
    a = eng.psis_loo(src, M, method, -2.0, 0.7, rows=idx)
    b = eng.psis_loo(sub, M, method, -2.0, 0.7)
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        x = a[key].cpu().numpy() if where == "device" else a[key]
        y = b[key].cpu().numpy() if where == "device" else b[key]
        np.testing.assert_array_equal(x, y, err_msg=key)  # same kernels on the same rows: bitwise
    wa, wb = eng.waic(src, 1.0, rows=idx), eng.waic(sub, 1.0)
    for key in ("lppd_i", "var_i", "waic_i", "agg"):
        x = wa[key].cpu().numpy() if where == "device" else wa[key]
        y = wb[key].cpu().numpy() if where == "device" else wb[key]
        np.testing.assert_array_equal(x, y, err_msg=key)


def test_row_selection_rejects_bad_indices(eng):
    ll = make_ll(9, 256, np.float64, 3)
    with pytest.raises(IndexError):
        eng.psis_loo(ll, 48, "psis", rows=np.array([0, 9]))
    with pytest.raises(IndexError):
        eng.waic(ll, rows=np.array([-1]))
    out = eng.psis_loo(ll, 48, "psis", rows=np.array([], dtype=np.int64))
    assert out["loo_i"].size == 0 and out["agg"][0] == 0


@pytest.mark.parametrize("estimator", ["diff_srs", "srs", "hh_pps"])
@pytest.mark.parametrize("approximation", ["lpd", "sis", "tis"])
def test_subsample_vs_oracle(estimator, approximation):
    import pyloo_amd as pl
    import importlib

    ls = importlib.import_module("pyloo_amd.loo_subsample")  # (the package attribute of that name is the function)

    N, S, m, reff = 300, 1000, 40, 0.8
    ll = make_ll(N, S, np.float64, 11)
    # expected, with the oracle doing every pass and the same draw of observations
    if approximation == "lpd":
        approx = np.array([orc.lse(r, b_inv=S) for r in ll])
    else:
        approx = orc.loo_pointwise(ll, 1.0, approximation)["loo_i"]
    np.random.seed(7)
    ind = ls.subsample_indices(estimator, approx, m)
    pw = orc.loo_pointwise(ll[ind.idx], reff)
    var = ll[ind.idx].var(axis=1)
    if estimator == "hh_pps":
        z = ls.compute_sampling_probabilities(approx)[ind.idx]
        e, p = ls.hansen_hurwitz_estimate(z, ind.m_i, pw["loo_i"], N), ls.hansen_hurwitz_estimate(z, ind.m_i, var, N)
    elif estimator == "srs":
        e, p = ls.srs_estimate(pw["loo_i"], N), ls.srs_estimate(var, N)
    else:
        e, p = ls.diff_srs_estimate(approx, pw["loo_i"], ind.idx), ls.srs_estimate(var, N)
    np.random.seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out, got_ind, _ = pl.loo_subsample_from_matrix(ll, m, approximation, estimator, reff=reff, pointwise=True)
    np.testing.assert_array_equal(got_ind.idx, ind.idx)
    np.testing.assert_allclose(out["elpd_loo"], e.y_hat, rtol=1e-9)
    np.testing.assert_allclose(out["se"], np.sqrt(e.hat_v_y), rtol=1e-7)
    np.testing.assert_allclose(out["subsampling_SE"], np.sqrt(e.v_y_hat), rtol=1e-7)
    np.testing.assert_allclose(out["p_loo"], p.y_hat, rtol=1e-9)
    np.testing.assert_allclose(out["p_loo_se"], np.sqrt(p.hat_v_y), rtol=1e-7)
    np.testing.assert_allclose(out["pareto_k"], pw["diag"], rtol=1e-9, atol=1e-10)
    full = out["loo_i"]
    assert np.isnan(full).sum() == N - len(ind.idx)
    np.testing.assert_allclose(full[ind.idx], pw["loo_i"], rtol=1e-9, atol=1e-10)
    assert out["subsample_size"] == len(ind.idx) and out["n_data_points"] == N and out["method"] == "loo_subsample"
    assert list(out.index[:8]) == ["elpd_loo", "se", "p_loo", "p_loo_se", "p_loo_subsampling_se", "n_samples",
                                   "n_data_points", "warning"]


def test_sampling_every_observation_reproduces_loo():
    import torch

    import pyloo_amd as pl

    N, S = 200, 2000
    ll = make_ll(N, S, np.float64, 5)
    t = torch.as_tensor(ll).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = pl.loo_from_matrix(t, reff=1.0, pointwise=True)
        for est in ("diff_srs", "srs"):
            out, _, _ = pl.loo_subsample_from_matrix(t, np.arange(N), "lpd", est, pointwise=True)
            np.testing.assert_allclose(out["elpd_loo"], ref["elpd_loo"], rtol=1e-12)
            np.testing.assert_array_equal(np.asarray(out["loo_i"]), ref["loo_i"].cpu().numpy())
            assert out["subsampling_SE"] == 0.0  # the whole population was observed
        scaled, _, _ = pl.loo_subsample_from_matrix(t, np.arange(N), "lpd", "srs", scale="deviance")
        np.testing.assert_allclose(scaled["elpd_loo"], -2 * ref["elpd_loo"], rtol=1e-12)


def test_importance_resample_draws_the_reference_indices():
    """``importance_resample`` (loo_approximate_posterior.py:437-536, used by the posterior correction of ``loo_subsample``)
    against index arrays the reference's function returned for the same ``log_p`` / ``log_q`` / method / seed
    (tests/golden/make_golden_resample.py): smoothing on the device, the generator calls the reference's."""
    from conftest import load_golden
    from pyloo_amd.loo_subsample import importance_resample

    g = load_golden("resample")
    for i, case in enumerate(g["cases"]):
        method, seed = str(case).split(":")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx = importance_resample(g[f"c{i}_log_p"], g[f"c{i}_log_q"], method=method, seed=int(seed))
        want = g[f"c{i}_idx"]
        assert idx.shape == want.shape
        # (the probabilities agree to ~1e-12 with the reference's: a draw changes only where a uniform number falls that close
        # to a boundary of the cumulative weights)
        assert np.mean(idx == want) >= 0.999, (case, np.mean(idx == want))
        if method != "psir":
            assert len(np.unique(idx)) == len(idx)  # without replacement: a permutation of the draws
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(Exception) as err:  # non-finite ratios: the reference's function ends in this exception as well
            importance_resample(g["nonfinite_log_p"], g["nonfinite_log_q"], method="psis", seed=17)
    assert type(err.value).__name__ == str(g["nonfinite_raises"])


@pytest.mark.parametrize("where", ["host", "device"])
def test_posterior_correction_re_draws_the_sampled_rows(where):
    """loo_subsample.py:333-370: with ``log_p`` / ``log_q`` the draws of the SAMPLED rows are re-drawn before PSIS-LOO and the
    variance over draws.  "psis" draws without replacement -- a permutation, which PSIS-LOO does not see; "psir" draws with
    replacement: against the oracle on the matrix gathered on the host."""
    import torch

    from pyloo_amd.loo_subsample import importance_resample, loo_subsample_from_matrix

    N, S, m = 300, 2000, 40
    ll = make_ll(N, S, np.float64, 99)
    rng = np.random.default_rng(5)
    log_q = rng.normal(size=S)
    log_p = log_q + rng.standard_t(df=5, size=S)
    obs = np.sort(rng.permutation(N)[:m])
    src = torch.as_tensor(ll).cuda() if where == "device" else ll
    base, _, _ = loo_subsample_from_matrix(src, obs, "lpd", "diff_srs", None, 1.0, None, True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        perm = importance_resample(log_p, log_q, method="psis", seed=3)
        boot = importance_resample(log_p, log_q, method="psir", seed=3)
        a, _, _ = loo_subsample_from_matrix(src, obs, "lpd", "diff_srs", None, 1.0, None, True, draw_index=perm)
        b, _, _ = loo_subsample_from_matrix(src, obs, "lpd", "diff_srs", None, 1.0, None, True, draw_index=boot)
    np.testing.assert_allclose(a["elpd_loo"], base["elpd_loo"], rtol=1e-9)
    np.testing.assert_allclose(a["p_loo"], base["p_loo"], rtol=1e-7)
    ref = orc.loo_arrays(ll[obs][:, boot], 1.0)
    got = np.asarray(b["loo_i"])
    assert np.count_nonzero(np.isfinite(got)) >= m
    np.testing.assert_allclose(got[obs], ref["loo_i"], rtol=1e-8, atol=1e-10)
