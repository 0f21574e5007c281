"""Host side of loo_subsample: the survey-sampling estimators and the subsample draw against
golden vectors produced by the reference's own ``pyloo/estimators`` modules (tests/golden/make_golden_estimators.py),
and the argument handling of the front.  No GPU."""

import importlib
import os

import numpy as np
import pytest

ls = importlib.import_module("pyloo_amd.loo_subsample")  # (the package attribute of that name is the function)

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "estimators.npz"))
CASES = [(0, 500, 40, 1), (1, 3000, 300, 2), (2, 64, 2, 3), (3, 1000, 1000, 4)]


@pytest.mark.parametrize("case,N,m,seed", CASES)
@pytest.mark.parametrize("est", ["diff_srs", "srs", "hh_pps"])
def test_estimators_match_reference(case, N, m, seed, est):
    p = f"c{case}_"
    approx, truth = GOLD[p + "approx"], GOLD[p + "truth"]
    np.random.seed(100 + seed)  # the reference draws from the global NumPy state (estimators/base.py:98-117)
    ind = ls.subsample_indices(est, approx, m)
    np.testing.assert_array_equal(ind.idx, GOLD[p + est + "_idx"])
    np.testing.assert_array_equal(ind.m_i, GOLD[p + est + "_m_i"])
    y = truth[ind.idx]
    if est == "diff_srs":
        r = ls.diff_srs_estimate(approx, y, ind.idx)
    elif est == "srs":
        r = ls.srs_estimate(y, N)
    else:
        z = ls.compute_sampling_probabilities(approx)
        np.testing.assert_allclose(z, GOLD[p + "z"], rtol=1e-15)
        r = ls.hansen_hurwitz_estimate(z[ind.idx], ind.m_i, y, N)
    got = np.array([float(v) for v in r])
    np.testing.assert_allclose(got, GOLD[p + est + "_result"], rtol=1e-12, atol=0.0)


def test_uniform_probabilities_when_all_zero():
    np.testing.assert_array_equal(ls.compute_sampling_probabilities(np.zeros(5)), GOLD["zero_probabilities"])


def test_estimator_input_checks():
    with pytest.raises(ValueError, match="same length"):
        ls.diff_srs_estimate(np.zeros(5), np.zeros(2), np.array([0, 1, 2]))
    with pytest.raises(ValueError, match="invalid indices"):
        ls.diff_srs_estimate(np.zeros(5), np.zeros(2), np.array([0, 7]))
    with pytest.raises(ValueError, match="must be positive"):
        ls.hansen_hurwitz_estimate(np.array([0.5, 0.0]), np.array([1, 1]), np.zeros(2), 10)
    with pytest.raises(ValueError, match="cannot exceed"):
        ls.subsample_indices("srs", np.zeros(5), 6)
    with pytest.raises(ValueError, match="Unknown estimator"):
        ls.subsample_indices("nope", np.zeros(5), 2)
    one = ls.diff_srs_estimate(np.arange(4.0), np.array([1.5]), np.array([2]))  # m = 1: no variance (difference.py:100-102)
    assert np.isinf(one.v_y_hat) and np.isinf(one.hat_v_y)


def test_front_argument_errors():
    ll = np.zeros((10, 64))
    with pytest.raises(ValueError, match="Invalid loo_approximation"):
        ls.loo_subsample_from_matrix(ll, 5, loo_approximation="exact")
    with pytest.raises(ValueError, match="Invalid estimator"):
        ls.loo_subsample_from_matrix(ll, 5, estimator="jackknife")
    with pytest.raises(ValueError, match="between 1 and 10"):
        ls.loo_subsample_from_matrix(ll, 11)
    with pytest.raises(ValueError, match="between 0 and 9"):
        ls.loo_subsample_from_matrix(ll, np.array([0, 10]))
    with pytest.raises(TypeError, match="must contain integers"):
        ls.loo_subsample_from_matrix(ll, np.array([0.5, 1.0]))
    with pytest.raises(TypeError, match="None, an integer"):
        ls.loo_subsample_from_matrix(ll, "all")
    with pytest.raises(TypeError, match="Valid scale values"):
        ls.loo_subsample_from_matrix(ll, 5, scale="bits")
