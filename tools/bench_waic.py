#!/usr/bin/env python3
"""Secondary benchmark: the WAIC pass (pl.waic -> pla_waic) on a device-resident matrix.

    python tools/bench_waic.py [--obs N] [--draws S] [--dtype f64|f32] [--steps K] [--warmup W]

Algorithmic bytes per observation: S*sizeof(T) read + 24 written.  One JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs", type=int, default=1_000_000)
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch

    from oracle import psis_oracle as orc
    from pyloo_amd.engine import get_engine

    eng = get_engine(0)
    S, N = args.draws, args.obs
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esz = 8 if args.dtype == "f64" else 4
    ll = torch.empty((N, S), dtype=tdt, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0003)
    for _ in range(args.warmup):
        res = eng.waic(ll, 1.0, pointwise=False)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = eng.waic(ll, 1.0, pointwise=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_ms()
    eng.set_timing(False)
    kernel_ms = k_ms / max(k_n, 1)
    alg = N * (S * esz + 24.0)
    idx = np.arange(0, N, max(N // 2048, 1))[:2048]
    rows = ll[idx].cpu().numpy().astype(np.float64)
    c0 = time.perf_counter()
    want = orc.waic_arrays(rows, 1)
    t_cpu = time.perf_counter() - c0
    got = eng.waic(ll, 1.0)["waic_i"][idx].cpu().numpy()
    err = float(np.max(np.abs(got - want["waic_i"]) / np.maximum(np.abs(want["waic_i"]), 1e-2)))
    print(json.dumps({
        "metric": "waic_observations_per_second", "value": N * args.steps / dt, "unit": "obs/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "dtype": args.dtype,
        "config": {"workload": f"WAIC, synthetic {args.dtype} S={S} x N={N}, device-resident"},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": {"value": len(idx) / t_cpu, "unit": "obs/s", "cores": 1, "kind": "port",
                         "sample": f"{len(idx)} strided rows, NumPy oracle (waic.py:137-161 restated)"},
        "parity": {"rows": int(len(idx)), "max_rel_err": {"waic_i": err}, "tolerance": 1e-6},
    }), flush=True)


if __name__ == "__main__":
    main()
