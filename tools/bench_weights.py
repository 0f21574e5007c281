#!/usr/bin/env python3
"""Secondary benchmark: the weights-returning flavour of the hot path (pl.psislw /
compute_importance_weights -> pla_importance_weights) on a device-resident matrix.

    python tools/bench_weights.py [--obs N] [--draws S] [--dtype f64|f32] [--steps K] [--warmup W]

Algorithmic bytes per observation: S*sizeof(T) read + S*sizeof(T) written + 8 (k-hat).  Prints one JSON
line; the headline metric of the repo stays bench.py."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs", type=int, default=500_000)
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default="psis", choices=["psis", "sis", "tis"])
    args = ap.parse_args()
    import numpy as np
    import torch

    from oracle import psis_oracle as orc
    from pyloo_amd.base import tail_count_for
    from pyloo_amd.engine import get_engine

    eng = get_engine(0)
    S, N = args.draws, args.obs
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esz = 8 if args.dtype == "f64" else 4
    ll = torch.empty((N, S), dtype=tdt, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0003)
    ll.neg_()  # log ratios of LOO: -log_lik
    M = tail_count_for(S, 1.0) if args.method == "psis" else 0
    for _ in range(args.warmup):
        lw, k = eng.importance_weights(ll, M, args.method)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lw, k = eng.importance_weights(ll, M, args.method)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_ms()
    eng.set_timing(False)
    kernel_ms = k_ms / max(k_n, 1)
    alg = N * (2.0 * S * esz + 8.0)
    # parity on a sample against the oracle (also a CPU baseline of this flavour)
    idx = np.arange(0, N, max(N // 256, 1))[:256]
    c0 = time.perf_counter()
    ref_lw, ref_k = orc.importance_weights(ll[idx].cpu().numpy().astype(np.float64), args.method, 1.0)
    t_cpu = time.perf_counter() - c0
    got = lw[idx].cpu().numpy().astype(np.float64)
    # tied tail draws (f32 rows have them now and then): the reference hands their quantiles out in the order of an unstable
    # argsort (psis.py:146), the engine in another: such rows hold the same MULTISET of weights and are compared sorted, as in
    # tests/test_gpu_parity.py
    row_err = np.max(np.abs(got - ref_lw) / np.maximum(np.abs(ref_lw), 1e-2), axis=1)
    tied = np.flatnonzero(row_err > 1e-6)
    for i in tied:
        row_err[i] = np.max(np.abs(np.sort(got[i]) - np.sort(ref_lw[i])) / np.maximum(np.abs(np.sort(ref_lw[i])), 1e-2))
    err_lw = float(np.max(row_err))
    err_k = float(np.max(np.abs(k[idx].cpu().numpy() - ref_k) / np.maximum(np.abs(ref_k), 1e-2)))
    print(json.dumps({
        "metric": f"{args.method}lw_observations_per_second" if args.method != "psis" else "psislw_observations_per_second", "value": N * args.steps / dt, "unit": "obs/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "dtype": args.dtype,
        "config": {"workload": f"{args.method} weights out, synthetic {args.dtype} S={S} x N={N}, reff=1 (M={M}), device-resident"},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": {"value": len(idx) / t_cpu, "unit": "obs/s", "cores": 1, "kind": "port",
                         "sample": f"{len(idx)} strided rows, NumPy oracle psislw"},
        "parity": {"rows": int(len(idx)), "max_rel_err": {"lw": err_lw, "khat": err_k}, "tolerance": 1e-6,
                   "rows_compared_as_multisets_of_weights": int(len(tied)), "output_dtype": args.dtype},
    }), flush=True)


if __name__ == "__main__":
    main()
