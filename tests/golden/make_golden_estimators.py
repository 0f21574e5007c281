#!/usr/bin/env python3
"""Golden vectors for the subsampling estimators of ``loo_subsample`` from the REAL reference
(``pyloo/estimators/*.py``, pure NumPy), loaded in place from ``/root/reference`` like make_golden.py does.
Run only in the build container:  ``python tests/golden/make_golden_estimators.py``  -> estimators.npz
Only inputs and the numbers the reference's functions return are written."""

import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pyloo/estimators"


def load_estimators():
    pkg = types.ModuleType("pyloo_ref_estimators")
    pkg.__path__ = [REF]
    sys.modules["pyloo_ref_estimators"] = pkg
    mods = {}
    for name in ("base", "difference", "hansen_hurwitz", "srs"):
        spec = importlib.util.spec_from_file_location(f"pyloo_ref_estimators.{name}", f"{REF}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"pyloo_ref_estimators.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def main():
    mods = load_estimators()
    out = {}
    fields = ("y_hat", "v_y_hat", "hat_v_y", "m", "N", "subsampling_SE")
    for case, (N, m, seed) in enumerate(((500, 40, 1), (3000, 300, 2), (64, 2, 3), (1000, 1000, 4))):
        rng = np.random.default_rng(seed)
        approx = -np.abs(rng.normal(1.5, 0.7, size=N)) - 0.05
        truth = approx + rng.normal(0, 0.05, size=N)
        p = f"c{case}_"
        out[p + "approx"], out[p + "truth"] = approx, truth
        for est in ("diff_srs", "srs", "hh_pps"):
            np.random.seed(100 + seed)
            ind = mods["base"].subsample_indices(est, approx, m)
            out[p + est + "_idx"], out[p + est + "_m_i"] = ind.idx, ind.m_i
            y = truth[ind.idx]
            if est == "diff_srs":
                r = mods["difference"].diff_srs_estimate(y, approx, ind.idx)
            elif est == "srs":
                r = mods["srs"].srs_estimate(y, N)
            else:
                z = mods["hansen_hurwitz"].compute_sampling_probabilities(approx)
                out[p + "z"] = z
                r = mods["hansen_hurwitz"].hansen_hurwitz_estimate(z[ind.idx], ind.m_i, y, N)
            out[p + est + "_result"] = np.array([float(getattr(r, f)) for f in fields])
    out["zero_probabilities"] = mods["hansen_hurwitz"].compute_sampling_probabilities(np.zeros(5))
    np.savez_compressed(os.path.join(HERE, "estimators.npz"), **out)
    print("wrote estimators.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
