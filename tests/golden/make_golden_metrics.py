#!/usr/bin/env python3
"""Golden vectors for ``loo_predictive_metric`` and ``loo_score`` from the REAL reference's primitives.

Run only in the build container:  ``python tests/golden/make_golden_metrics.py``  -> metrics.npz

``pyloo/loo_predictive_metric.py`` is loaded in place from ``/root/reference`` (loader of make_golden.py / make_golden_e_loo.py:
its ``from arviz import InferenceData`` binds to the empty placeholder).  Its five reducers -- ``_mae``, ``_mse``, ``_rmse``,
``_accuracy``, ``_balanced_accuracy`` (234-356) -- are pure NumPy and run as they are.  The body of ``loo_predictive_metric``
and all of ``loo_score`` walk xarray objects and are not importable here; what they compute per observation is chained from
the reference's own pieces instead: ``psis.psislw`` (208 / loo_score.py:227, 311), the weighted mean of e_loo.py:437 with
``utils._logsumexp`` (557-559), NumPy's global ``permutation`` where loo_score.py:305 draws it; the two lines of ``_crps``
(loo_score.py:343-346) are evaluated inline.  Only inputs and the resulting numbers are written."""

import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_e_loo as mge  # noqa: E402

METRICS = ("mae", "mse", "rmse", "acc", "balanced_acc")


def load_metric_module():
    mods, _ = mge.load_e_loo()
    spec = importlib.util.spec_from_file_location("pyloo.loo_predictive_metric", f"{mg.REF}/loo_predictive_metric.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["pyloo.loo_predictive_metric"] = m
    spec.loader.exec_module(m)
    return mods, m


def weighted_mean(mods, x, lw):
    w = np.exp(lw - mods["utils"]._logsumexp(lw, axis=-1, keepdims=True))  # e_loo.py:557-559, 434
    return (w * x).sum(axis=-1)                                            # 437


def main():
    mods, pm = load_metric_module()
    psislw = mods["psis"].psislw
    reducers = {"mae": pm._mae, "mse": pm._mse, "rmse": pm._rmse, "acc": pm._accuracy, "balanced_acc": pm._balanced_accuracy}
    rng = np.random.default_rng(20261004)
    out = {}
    # 1. the reducers on their own
    y = rng.normal(size=200)
    yhat = y + rng.normal(scale=0.7, size=200)
    yb = (rng.uniform(size=200) < 0.4).astype(float)
    pb = np.clip(0.6 * yb + 0.2 + rng.normal(scale=0.2, size=200), 0.0, 1.0)
    out["red_y"], out["red_yhat"], out["red_yb"], out["red_pb"] = y, yhat, yb, pb
    for name, f in reducers.items():
        r = f(yb, pb) if name in ("acc", "balanced_acc") else f(y, yhat)
        out[f"red_{name}"] = np.array([r["estimate"], r["se"]])
    # 2. the chain of loo_predictive_metric (208-231): continuous and binary predictions
    n, s, reff = 24, 500, 0.8
    theta = rng.normal(size=(1, s)) * 0.5
    yobs = rng.normal(size=n) * 1.2
    ll = -0.5 * (yobs[:, None] - theta) ** 2 - 0.9189385332046727
    x = theta + rng.normal(size=(n, s))                    # posterior-predictive draws
    lw, _ = psislw(-ll, reff=reff)
    pred = weighted_mean(mods, x, lw)
    out["pm_x"], out["pm_ll"], out["pm_y"], out["pm_reff"], out["pm_pred"] = x, ll, yobs, np.array(reff), pred
    for name in ("mae", "mse", "rmse"):
        r = reducers[name](yobs, pred)
        out[f"pm_{name}"] = np.array([r["estimate"], r["se"]])
    ybin = (rng.uniform(size=n) < 0.5).astype(float)
    p = 1.0 / (1.0 + np.exp(-(theta * 2.0 + (2 * ybin[:, None] - 1) * rng.uniform(-0.2, 0.9, size=(n, 1)) + rng.normal(size=(n, s)) * 0.3)))
    llb = ybin[:, None] * np.log(p) + (1 - ybin[:, None]) * np.log1p(-p)
    lwb, _ = psislw(-llb, reff=1.0)
    predb = weighted_mean(mods, p, lwb)
    out["pmb_x"], out["pmb_ll"], out["pmb_y"], out["pmb_pred"] = p, llb, ybin, predb
    for name in ("acc", "balanced_acc"):
        r = reducers[name](ybin, predb)
        out[f"pmb_{name}"] = np.array([r["estimate"], r["se"]])
    # 3. the chain of loo_score (219-239, 277-323), two permutations, CRPS and SCRPS
    x2 = theta + rng.normal(size=(n, s))
    out["sc_x2"] = x2
    for scale in (False, True):
        np.random.seed(1234)
        exx = 0.0
        for _ in range(2):
            shuffle = np.random.permutation(s)                                     # 305
            joint = -ll - ll[:, shuffle]                                           # 310
            lwj, _ = psislw(joint, reff=reff)                                      # 311
            exx = exx + weighted_mean(mods, np.abs(x - x2[:, shuffle]), lwj)      # 313-320
        exx = exx / 2                                                              # 225
        lw1, k1 = psislw(-ll, reff=reff)                                           # 227
        exy = weighted_mean(mods, np.abs(x - yobs[:, None]), lw1)                  # 230-237
        score = (-exy / exx - 0.5 * np.log(exx)) if scale else (0.5 * exx - exy)  # 343-346
        tag = "scrps" if scale else "crps"
        out[f"sc_{tag}_pw"] = score
        out[f"sc_{tag}_est"] = np.array([score.mean(), score.std() / np.sqrt(score.size)])  # 241-242
        out["sc_k"] = k1
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("wrote metrics.npz:", len(out), "arrays")
    for k in sorted(out):
        if out[k].size <= 2:
            print(k, out[k])


if __name__ == "__main__":
    main()
