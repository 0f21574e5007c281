#!/usr/bin/env python3
"""Golden vectors for the weighted expectations of ``e_loo`` from the REAL reference.

Run only in the build container:  ``python tests/golden/make_golden_e_loo.py``  -> e_loo.npz

``pyloo/e_loo.py`` is loaded in place from ``/root/reference`` with the loader of make_golden.py (its module-level
``import xarray`` / ``from arviz import InferenceData`` bind to the same two empty placeholder modules).  The per-row
NumPy helpers run as they are: ``k_hat`` (e_loo.py:328-390), ``_wvar_func`` (518-531), ``_weighted_quantile`` (534-554),
``_pareto_min_ss`` / ``_pareto_khat_threshold`` / ``_pareto_convergence_rate`` (393-427) and ``utils._logsumexp`` for
``_normalize_log_weights`` (557-559).  The xarray wrappers around them (``_compute_weighted_mean`` 430-437 etc.) need real
xarray and are NOT importable here: the one line each adds -- ``(weights * x).sum(dim="__sample__")`` -- is evaluated with
NumPy on the last axis.  Only inputs and the numbers the reference's functions return are written."""

import importlib.util
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

PROBS = np.array([0.05, 0.5, 0.9])


def load_e_loo():
    mods = mg.load_reference()
    spec = importlib.util.spec_from_file_location("pyloo.e_loo", f"{mg.REF}/e_loo.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["pyloo.e_loo"] = m
    spec.loader.exec_module(m)
    return mods, m


def rows_case(rng, n, s, dtype=np.float64):
    """Smoothed log-weights from the reference's own psislw on heavy-ish log ratios, and draws x to average."""
    k = rng.uniform(0.1, 0.9, size=(n, 1))
    lr = (k * rng.exponential(size=(n, s))).astype(dtype)
    x = (rng.normal(size=(n, s)) * rng.uniform(0.5, 3.0, size=(n, 1)) + rng.normal(size=(n, 1))).astype(dtype)
    return x, lr


def reference_rows(mods, el, x, lw, lr):
    n, s = x.shape
    nlw = lw - mods["utils"]._logsumexp(lw, axis=-1, keepdims=True)  # e_loo.py:557-559
    w = np.exp(nlw)
    out = {
        "mean": (w * x).sum(axis=-1),  # e_loo.py:437
        "var": np.array([el._wvar_func(x[i], w[i]) for i in range(n)], dtype=np.float64),
        "quant": np.array([[el._weighted_quantile(x[i], w[i], p) for p in PROBS] for i in range(n)], dtype=np.float64),
        "k_mean": np.array([el.k_hat(x[i], lr[i]) for i in range(n)], dtype=np.float64),
        "k_var": np.array([el.k_hat(x[i] ** 2, lr[i]) for i in range(n)], dtype=np.float64),  # e_loo.py:232-234
        "k_none": np.array([el.k_hat(None, lr[i]) for i in range(n)], dtype=np.float64),  # quantiles: e_loo.py:230
    }
    return out


def main():
    warnings.simplefilter("ignore")
    mods, el = load_e_loo()
    out = {"probs": PROBS}
    rng = np.random.default_rng(20261004)
    cases = {}
    # ordinary rows
    for name, (n, s) in {"s4000": (5, 4000), "s1000": (6, 1000), "s257": (8, 257), "s64": (8, 64), "s16": (6, 16), "s4": (5, 4)}.items():
        x, lr = rows_case(rng, n, s)
        lw = mods["psis"].psislw(-(-lr), 1.0)[0] if s >= 8 else lr - mods["utils"]._logsumexp(lr, axis=-1, keepdims=True)
        cases[name] = (x, lw, lr)
    # edge rows, S = 500: every branch of k_hat / _wvar_func / _weighted_quantile
    n, s = 14, 500
    x, lr = rows_case(rng, n, s)
    lw = mods["psis"].psislw(lr, 1.0)[0]
    x[0] = 2.5                                  # constant x: allclose(x, x[0]) (e_loo.py:357, 520)
    x[1] = np.where(rng.random(s) < 0.3, 1.0, 0.0)  # binary outcomes: exactly two unique values (358)
    x[2, 17] = np.nan                           # NaN in x (359)
    x[3, 5] = np.inf                            # inf in x (360)
    lw[4] = -np.log(s)                          # constant weights: np.quantile branch (536), w_sum_sq = 1/S
    lr[4] = 0.0                                 # constant ratios: khat_r = inf (347)
    lr[5, :30] = lr[5].max() + 1.0              # 30 tied largest ratios: top-20 allclose -> inf (347)
    lw[6] = -800.0
    lw[6, 3] = 0.0                              # one draw carries all the weight: isclose(w_sum_sq, 1) -> 0 (524)
    x[7] = np.round(x[7], 1)                    # many ties in x (weighted quantile with equal neighbours)
    x[8] = np.abs(x[8]) + 1.0                   # strictly positive h
    x[9] = -np.abs(x[9]) - 1.0                  # strictly negative h
    x[10, :25] = x[10].max() + 5.0              # right tail of h*r allclose only if the ratios agree too: usually not
    lr[10, :25] = lr[10].max()
    x[11] = x[11] * 1e-9 + 3.0                  # allclose(x, x[0]) through the relative tolerance
    x[12, ::2] = 1.0
    x[12, 1::2] = 1.0 + 1e-7                    # two unique values that are also allclose
    lr[13, 100] = np.nan                        # NaN in the log ratios
    cases["edges_s500"] = (x, lw, lr)
    # f32 inputs (the reference keeps the input dtype through exp / sort)
    x, lr = rows_case(rng, 6, 1000, np.float32)
    cases["s1000_f32"] = (x, mods["psis"].psislw(lr, 1.0)[0], lr)
    # draws with many equal values (count / binary / rounded posterior-predictive data): the quantile is crossed INSIDE a group of
    # equal draws in most rows, where the reference returns the value itself (e_loo.py:548-554 with x1 == x_sorted[wi]).  A
    # generator of its own, so that the cases above keep their numbers.
    rng2 = np.random.default_rng(20261005)
    n, s = 10, 2000
    x, lr = rows_case(rng2, n, s)
    lw = mods["psis"].psislw(lr, 1.0)[0]
    x[0] = rng2.poisson(5.0, s)
    x[1] = rng2.poisson(0.5, s)
    x[2] = rng2.binomial(1, 0.3, s)
    x[3] = rng2.binomial(12, 0.4, s)
    x[4] = np.round(x[4], 0)
    x[5] = np.round(x[5], 1)
    x[6] = np.floor(np.abs(x[6]) * 3.0)
    x[7] = rng2.poisson(40.0, s)
    x[8] = np.where(rng2.random(s) < 0.9, 0.0, x[8])  # zero-inflated
    x[9] = np.sign(x[9])
    cases["ties_s2000"] = (x, lw, lr)
    for name, (x, lw, lr) in cases.items():
        ref = reference_rows(mods, el, x, lw, lr)
        out[name + "_x"], out[name + "_lw"], out[name + "_lr"] = x, lw, lr
        for key, v in ref.items():
            out[f"{name}_{key}"] = v
    # the three scalar diagnostics (e_loo.py:393-427)
    ks = np.array([-0.3, 0.0, 0.1, 1 / 6, 0.49, 0.5, 0.51, 0.7, 0.99, 1.0, 1.2, np.inf, -np.inf, np.nan])
    out["diag_k"] = ks
    out["diag_min_ss"] = np.array([el._pareto_min_ss(k) for k in ks])
    for s in (16, 500, 4000):
        out[f"diag_threshold_{s}"] = np.array(el._pareto_khat_threshold(s))
        out[f"diag_rate_{s}"] = np.array([el._pareto_convergence_rate(k, s) for k in ks])
    np.savez_compressed(os.path.join(HERE, "e_loo.npz"), **out)
    print("wrote e_loo.npz:", len(out), "arrays;", {k: v[0].shape for k, v in cases.items()})
    for name in cases:
        print(name, "k_mean", np.unique(out[name + "_k_mean"]), "k_var", np.unique(out[name + "_k_var"]))


if __name__ == "__main__":
    main()
