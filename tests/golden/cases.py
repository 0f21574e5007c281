"""Closed-form (RNG-free) log-likelihood matrices for the golden fixtures.

Everything here is this repository's own test-input generator: it only decides
WHAT rows are fed to the reference; the expected outputs stored next to it in
``*.npz`` come from the reference's real functions (see ``make_golden.py``).

Row families follow SURVEY.md section 8c: generalised-Pareto shaped rows with
known tail index, Gaussian shaped rows, bounded weights (negative k), and the
in-band edge cases the reference handles (constant rows, tails of <= 4, ties
at the cutoff, +/-inf and NaN entries, rows spanning > 709 nats so the
``log(DBL_MIN)`` floor of ``psis.py:136`` binds, +/-1e10 extremes).
"""

import numpy as np
from scipy.special import ndtri

PERM_A = 1103  # prime, coprime with every S used below
PERM_B = 1619  # second prime for the Gaussian rows


def _perm(S, mult):
    return (np.arange(S, dtype=np.int64) * mult) % S


def exp_scores(S, mult=PERM_A):
    """Permuted exact quantiles of Exp(1): E_s = -log1p(-(pi(s)+0.5)/S)."""
    return -np.log1p(-(_perm(S, mult) + 0.5) / S)


def pareto_rows(S, ks):
    """ll[i, s] = -k_i * E_s + c_i, so the importance ratios have tail index ~k_i."""
    E = exp_scores(S)
    ks = np.asarray(ks, dtype=np.float64)
    c = -1.0 - (np.arange(len(ks)) % 7) * 0.25
    return -ks[:, None] * E[None, :] + c[:, None]


def gauss_rows(S, scales):
    z = ndtri((_perm(S, PERM_B) + 0.5) / S)
    return np.stack([-0.5 * (a * z) ** 2 for a in scales])


def bounded_row(S):
    """Ratios 1/(1+u) in (1/2, 1): bounded weights, negative tail index."""
    u = (_perm(S, PERM_A) + 0.5) / S
    return np.log1p(u)[None, :]


def edge_rows(S):
    """Rows for the in-band special cases. Returns (matrix, labels)."""
    E = exp_scores(S)
    rows, labels = [], []

    rows.append(np.full(S, -1.25))
    labels.append("constant")

    r = np.full(S, -5.0)
    r[[1 % S, (S // 2), S - 1]] = -7.0
    rows.append(r)
    labels.append("tail_le_4")

    rows.append(-np.round(0.5 * E * 8.0) / 8.0)
    labels.append("ties_grid")

    r = -0.4 * E - 2.0
    r[3 % S] = np.inf
    rows.append(r)
    labels.append("ll_pos_inf")

    r = -0.4 * E - 2.0
    r[5 % S] = -np.inf
    rows.append(r)
    labels.append("ll_neg_inf")

    r = -0.4 * E - 2.0
    r[2 % S] = np.nan
    rows.append(r)
    labels.append("ll_nan")

    rows.append(-300.0 * E)
    labels.append("floor_binds")

    r = -0.3 * E - 1.0
    r[0] = 1e10
    r[1] = -1e10
    rows.append(r)
    labels.append("extreme_1e10")

    r = -0.3 * E - 1.0
    r[S // 3] = 10.0
    rows.append(r)
    labels.append("one_large_ll")

    return np.stack(rows), labels


K_GRID = [0.05, 0.1, 0.3, 0.5, 0.7, 0.9, 1.2, 1.3]


def standard_matrix(S, dtype=np.float64, with_edges=True):
    parts = [pareto_rows(S, K_GRID), gauss_rows(S, [1.0, 3.0]), bounded_row(S)]
    labels = [f"pareto_k{k}" for k in K_GRID] + ["gauss_1", "gauss_3", "bounded"]
    if with_edges:
        e, el = edge_rows(S)
        parts.append(e)
        labels += el
    with np.errstate(all="ignore"):
        ll = np.concatenate(parts, axis=0).astype(dtype)
    return ll, labels


# (name, S, reff, dtype, with_edges)
CASES = [
    ("s8_r1_f64", 8, 1.0, "f64", True),
    ("s100_r0p3_f64", 100, 0.3, "f64", True),
    ("s100_r0p7_f64", 100, 0.7, "f64", True),
    ("s100_r1_f64", 100, 1.0, "f64", True),
    ("s100_r2_f64", 100, 2.0, "f64", True),
    ("s100_r1_f32", 100, 1.0, "f32", True),
    ("s2000_r1_f64", 2000, 1.0, "f64", True),
    ("s2000_r1_f32", 2000, 1.0, "f32", True),
    ("s4000_r0p3_f64", 4000, 0.3, "f64", False),
    ("s4000_r0p7_f64", 4000, 0.7, "f64", False),
    ("s4000_r1_f64", 4000, 1.0, "f64", True),
    ("s4000_r2_f64", 4000, 2.0, "f64", False),
    ("s4000_r1_f32", 4000, 1.0, "f32", True),
    ("s20000_r1_f64", 20000, 1.0, "f64", False),
    ("s20000_r1_f32", 20000, 1.0, "f32", False),
]

DTYPES = {"f64": np.float64, "f32": np.float32}


def known_answer_matrix():
    """SURVEY.md section 8c known-answer input: S=4000, k=0.1..1.2, c_i=0."""
    E = exp_scores(4000)
    ks = np.array([0.1, 0.3, 0.5, 0.7, 0.9, 1.2])
    return -ks[:, None] * E[None, :]
