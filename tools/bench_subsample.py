#!/usr/bin/env python3
"""Secondary benchmark: subsampled LOO (pl.loo_subsample_from_matrix) on a device-resident matrix.

    python tools/bench_subsample.py [--obs N] [--draws S] [--sample m] [--steps K] [--warmup W]

One step = the approximation pass over all N rows (SIS kernel, lppd_i = "lpd") + the draw of m observations + PSIS-LOO and
the variance over draws on the m sampled rows read in place (pla_psis_loo_rows / pla_waic_rows) + the difference estimator.
Algorithmic bytes: N*S*sizeof(T) for the approximation + 2*m*S*sizeof(T) for the sampled rows.  One JSON line, with the
elpd estimate next to the full-data loo() of the same matrix."""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs", type=int, default=1_000_000)
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--sample", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch

    import pyloo_amd as pl
    from pyloo_amd.engine import get_engine

    eng = get_engine(0)
    N, S, m = args.obs, args.draws, args.sample
    ll = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0003)
    warnings.simplefilter("ignore")
    np.random.seed(1)
    for _ in range(args.warmup):
        out, _, _ = pl.loo_subsample_from_matrix(ll, m, "lpd", "diff_srs")
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, ind, est = pl.loo_subsample_from_matrix(ll, m, "lpd", "diff_srs")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    k_ms, _ = eng.kernel_ms()
    eng.set_timing(False)
    c0 = time.perf_counter()
    np.random.choice(N, size=m, replace=False)  # the reference's draw (estimators/base.py:112-116) permutes all N indices
    t_draw = time.perf_counter() - c0
    full = pl.loo_from_matrix(ll)
    alg = (N + 2.0 * m) * S * 8.0
    print(json.dumps({
        "metric": "loo_subsample_ms_per_call", "value": dt * 1e3, "unit": "ms", "higher_is_better": False, "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "dtype": "f64",
        "config": {"workload": f"loo_subsample(lpd, diff_srs), synthetic f64 S={S} x N={N}, m={m} sampled rows, device-resident"},
        "algorithmic_gb_per_s": alg / dt / 1e9,
        "gpu_kernel_ms_per_call": k_ms / args.steps, "host_draw_of_the_subsample_ms": t_draw * 1e3,
        "elpd_loo_subsample": float(out["elpd_loo"]), "se": float(out["se"]), "subsampling_SE": float(out["subsampling_SE"]),
        "elpd_loo_full": float(full["elpd_loo"]),
        "z_score_of_the_difference": float((out["elpd_loo"] - full["elpd_loo"]) / max(out["subsampling_SE"], 1e-300)),
    }))


if __name__ == "__main__":
    main()
