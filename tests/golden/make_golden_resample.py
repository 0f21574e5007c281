#!/usr/bin/env python3
"""Golden vectors for the posterior correction of ``loo_subsample`` (log_p / log_q: loo_subsample.py:333-370) from the REAL
reference.  Run only in the build container:  ``python tests/golden/make_golden_resample.py``  -> resample.npz

``importance_resample`` (loo_approximate_posterior.py:437-536) lives in a module whose imports (arviz, xarray wrappers) are
not available here, so the FUNCTION is taken out of the reference's file as it lies there (its source text is read, parsed and
executed at run time -- nothing of it is written anywhere) and runs with the reference's own ``psislw`` and ``_logsumexp``
(loaded in place by make_golden.py).  Only inputs and the index arrays the function returns are written."""

import ast
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def load_function():
    mods = mg.load_reference()
    path = f"{mg.REF}/loo_approximate_posterior.py"
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "importance_resample")
    ns = {"np": np, "warnings": warnings, "psislw": mods["psis"].psislw, "_logsumexp": mods["utils"]._logsumexp}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["importance_resample"]


def main():
    f = load_function()
    rng = np.random.default_rng(20260105)
    out = {}
    S = 2000
    cases = []
    for i, (spread, method, seed) in enumerate([(0.3, "psis", 11), (1.0, "psis", 12), (1.0, "psir", 13), (2.5, "psir", 14),
                                                (1.0, "sis", 15), (3.0, "psis", 16)]):
        log_q = rng.normal(size=S) - 3.0
        log_p = log_q + spread * rng.standard_t(df=5, size=S)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx = f(log_p, log_q, method=method, seed=seed)
        out[f"c{i}_log_p"] = log_p
        out[f"c{i}_log_q"] = log_q
        out[f"c{i}_idx"] = np.asarray(idx, dtype=np.int64)
        cases.append(f"{method}:{seed}")
    out["cases"] = np.array(cases)
    # non-finite ratios: the reference drops them from the weights but still draws from all `draws` positions, numpy refuses
    # (a and p differ in length), the fallback draws unweighted indices up to draws - 1 and the map back to the original
    # positions runs out of bounds: the function ends in an exception, which loo_subsample turns into its warning
    # "Importance resampling failed ... Falling back to original samples".  Recorded as such.
    log_q = rng.normal(size=S)
    log_p = log_q + rng.normal(size=S)
    log_p[[3, 77, 1500]] = [np.inf, -np.inf, np.nan]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            f(log_p, log_q, method="psis", seed=17)
        raised = "none"
    except Exception as e:  # noqa: BLE001
        raised = type(e).__name__
    out["nonfinite_log_p"] = log_p
    out["nonfinite_log_q"] = log_q
    out["nonfinite_raises"] = np.array(raised)
    print("non-finite ratios:", raised)
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)
    for i, c in enumerate(cases):
        idx = out[f"c{i}_idx"]
        print(c, "indices", idx.shape, "distinct", len(np.unique(idx)), "first", idx[:6])


if __name__ == "__main__":
    main()
