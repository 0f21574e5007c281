"""Pin the CPU oracle (oracle/psis_oracle.py) to the reference's real outputs.

The fixtures under tests/golden/ were written by tests/golden/make_golden.py from the
reference's own functions (pyloo/psis.py, utils.py, sis.py, tis.py).  These tests run on
CPU only and never read /root/reference.
"""

import numpy as np
import pytest

import cases
from conftest import load_golden
from oracle import psis_oracle as orc

RT = 1e-12  # the oracle restates the same NumPy arithmetic: expect ~ulp agreement


def _close(a, b, rtol=RT, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


@pytest.mark.parametrize("case", [c[0] for c in cases.CASES])
def test_case_matches_reference(case):
    g = load_golden(case)
    ll = g["ll"]
    reff = float(g["reff"])
    ll64 = ll.astype(np.float64)
    res = orc.loo_pointwise(ll64, reff)
    _close(res["diag"], g["khat"])
    _close(res["lw"], g["lw"])
    _close(res["loo_i"], g["loo_i"])
    _close(res["lppd_i"], g["lppd_i"])
    ok = g["agg_rows"]
    agg = orc.loo_aggregate(res["loo_i"][ok], res["lppd_i"][ok], res["diag"][ok], ll.shape[1])
    for key in ("elpd_loo", "se", "lppd", "p_loo", "p_loo_se", "looic", "looic_se", "good_k"):
        _close(agg[key], g[key], rtol=1e-11)
    assert agg["n_high_k"] == int(g["n_high_k"])
    assert orc.tail_count(ll.shape[1], reff) == -int(g["cutoff_ind"]) - 1
    if ll.dtype == np.float32:  # the reference's own mixed-precision behaviour for f32 input
        nat = orc.loo_pointwise(ll, reff)
        assert nat["lw"].dtype == np.float32 and nat["diag"].dtype == np.float64
        _close(nat["diag"], g["native32_khat"])
        _close(nat["lw"], g["native32_lw"], rtol=1e-6)
        _close(nat["loo_i"], g["native32_loo_i"], rtol=1e-6)
        _close(nat["lppd_i"], g["native32_lppd_i"], rtol=1e-6)


def test_intermediates():
    g = load_golden("s4000_r1_f64")
    M = orc.tail_count(4000, 1.0)
    assert M == 190
    for i in range(g["ll"].shape[0]):
        d = {}
        with np.errstate(all="ignore"):
            orc.psis_row(-g["ll"][i], M, details=d)
        _close(d["xcutoff"], g["xcutoff"][i])
        assert d["tail_len"] == g["tail_len"][i]
        _close(d["sigma"], g["sigma"][i])


def test_known_answer():
    """SURVEY.md section 8c / BASELINE.md section 4 known-answer (real reference run)."""
    g = load_golden("known_answer_s4000")
    out = orc.loo_arrays(g["ll"], 1.0)
    _close(out["khat"], [0.1263015392349325, 0.31231166382304587, 0.4983134530821454,
                         0.6843220218846074, 0.8703208341622388, 1.1492715840633747], rtol=1e-12)
    _close(out["elpd_loo"], -7.495953310386053)
    _close(out["se"], 2.6906896870466506)
    _close(out["p_loo"], 4.771879530745729)
    _close(out["p_loo_se"], 1.0984694649055418)
    _close(out["lppd"], -2.7240737796403236)
    _close(out["khat"], g["khat"])
    _close(out["loo_i"], g["loo_i"])


def test_shapes_and_reference_edge_tests():
    g = load_golden("shapes")
    lw, k = orc.psislw(g["x1"], 0.7)
    assert isinstance(k, np.ndarray) and k.shape == ()  # test_psis.py:49-58
    _close(lw, g["lw1"])
    _close(k, g["k1"])
    lw, k = orc.psislw(g["x3"], 0.7)
    assert lw.shape == (2, 3, 100) and k.shape == (2, 3)
    _close(lw, g["lw3"])
    _close(k, g["k3"])
    lw, k = orc.psislw(g["small"])  # test_psis.py:95-99
    assert k == np.inf
    _close(lw, g["lw_small"])
    lw, k = orc.psislw(g["const"])  # test_psis.py:121-125
    assert k == np.inf
    _close(lw, -np.log(100.0) * np.ones(100))
    x = g["x1"].copy()
    orc.psislw(x)
    assert np.array_equal(x, g["x1"])  # inputs are never mutated (psis.py:78)


def test_unit_primitives():
    u = load_golden("units")
    i = 0
    while f"gpdfit_in_{i}" in u:
        k, s = orc.gpd_fit(u[f"gpdfit_in_{i}"])
        _close([k, s], u[f"gpdfit_out_{i}"])
        i += 1
    assert i == 7
    with np.errstate(all="ignore"):
        for row in u["gpinv_table"]:
            p, kappa, sigma, want = row[:3], row[3], row[4], row[5:]
            _close(orc.gpd_quantile(p, kappa, sigma), want)
    v = u["lse_in"]
    _close([orc.lse(r) for r in v], u["lse_plain"])
    _close([orc.lse(r, b_inv=100) for r in v], u["lse_binv"])
    got32 = [orc.lse(r) for r in v.astype(np.float32)]
    assert all(isinstance(x, np.float32) for x in got32)
    _close(got32, u["lse_f32"], rtol=1e-6)
    assert orc.lse(v[0], b_inv=0) == np.inf
    for nm, f in (("sis", orc.sis_row), ("tis", orc.tis_row)):
        for j, r in enumerate(-v):
            lw, ess = f(r)
            _close(lw, u[f"{nm}_lw"][j])
            _close(ess, u[f"{nm}_ess"][j])


def test_vectorised_backend_equals_the_loop():
    """oracle.loo_pointwise_vectorised (second CPU line of the bench) against the per-observation loop."""
    rng = np.random.default_rng(8)
    ll = -rng.uniform(0.05, 1.3, size=(96, 1)) * rng.exponential(size=(96, 2000)) - 1.0
    ll[3, 7] = np.nan
    ll[5] = -2.0
    ll[9, :300] = np.round(ll[9, :300] * 4) / 4
    ll[11, 0] = -np.inf
    a = orc.loo_pointwise(ll, 0.8)
    b = orc.loo_pointwise_vectorised(ll, 0.8, chunk=32)
    for key in ("diag", "loo_i", "lppd_i"):
        assert np.array_equal(np.isnan(a[key]), np.isnan(b[key])), key
        ok = np.isfinite(a[key])
        np.testing.assert_allclose(b[key][ok], a[key][ok], rtol=1e-12, atol=1e-13)


# ---------------------------------------------------------------------------------------------
# e_loo: weighted expectations and their k (fixtures: tests/golden/make_golden_e_loo.py, the reference's own helpers)
# ---------------------------------------------------------------------------------------------
E_LOO_CASES = ["s4000", "s1000", "s257", "s64", "s16", "s4", "edges_s500", "s1000_f32", "ties_s2000"]


@pytest.mark.parametrize("case", E_LOO_CASES)
def test_e_loo_rows(case):
    g = load_golden("e_loo")
    x, lw, lr = (g[f"{case}_{k}"] for k in ("x", "lw", "lr"))
    got = orc.e_loo_arrays(x, lw, lr, probs=g["probs"])
    for key in ("mean", "var", "quant", "k_mean", "k_var", "k_none"):
        want = g[f"{case}_{key}"]
        assert np.array_equal(np.isnan(got[key]), np.isnan(want)), key
        ok = ~np.isnan(want)
        np.testing.assert_allclose(got[key][ok], want[ok], rtol=1e-12, atol=0, err_msg=f"{case} {key}")


def test_e_loo_scalar_diagnostics():
    g = load_golden("e_loo")
    ks = g["diag_k"]
    np.testing.assert_array_equal(np.array([orc.pareto_min_ss(k) for k in ks]), g["diag_min_ss"])
    for s in (16, 500, 4000):
        assert orc.pareto_khat_threshold(s) == float(g[f"diag_threshold_{s}"])
        np.testing.assert_array_equal(np.array([orc.pareto_convergence_rate(k, s) for k in ks]), g[f"diag_rate_{s}"])
