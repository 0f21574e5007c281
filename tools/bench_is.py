#!/usr/bin/env python3
"""Secondary benchmark: SIS / TIS LOO pass (method="sis" | "tis") on a device-resident matrix.
    python tools/bench_is.py [--method tis] [--obs N] [--draws S]      One JSON line."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--method", default="tis", choices=["sis", "tis"])
    ap.add_argument("--obs", type=int, default=1_000_000)
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    from oracle import psis_oracle as orc
    from pyloo_amd.engine import get_engine

    eng = get_engine(0)
    S, N = args.draws, args.obs
    ll = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(ll, seed=0x5EED0003, k_lo=0.05, k_hi=1.2)
    for _ in range(args.warmup):
        res = eng.psis_loo(ll, 0, args.method, 1.0, 0.7, pointwise=False)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = eng.psis_loo(ll, 0, args.method, 1.0, 0.7, pointwise=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_ms()
    eng.set_timing(False)
    kernel_ms = k_ms / max(k_n, 1)
    alg = N * (S * 8 + 24.0)
    idx = np.arange(0, N, max(N // 1024, 1))[:1024]
    c0 = time.perf_counter()
    ref = orc.loo_pointwise(ll[idx].cpu().numpy(), 1.0, args.method)
    t_cpu = time.perf_counter() - c0
    got = eng.psis_loo(ll, 0, args.method, 1.0, 0.7)
    err = {k2: float(np.max(np.abs(got[k1][idx].cpu().numpy() - ref[k2]) / np.maximum(np.abs(ref[k2]), 1e-2)))
           for k1, k2 in (("diag", "diag"), ("loo_i", "loo_i"), ("lppd_i", "lppd_i"))}
    print(json.dumps({
        "metric": f"{args.method}_loo_observations_per_second", "value": N * args.steps / dt, "unit": "obs/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "dtype": "f64",
        "config": {"workload": f"{args.method.upper()}-LOO, synthetic f64 S={S} x N={N}, device-resident",
                   "rows_for_general_kernel": int(got["agg"][7].item())},
        "roofline": {"bound": "hbm", "achieved": alg / (kernel_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": alg / (kernel_ms * 1e-3) / 1e9 / 8000.0, "traffic": None, "kernel_ms": kernel_ms,
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": {"value": len(idx) / t_cpu, "unit": "obs/s", "cores": 1, "kind": "port",
                         "sample": f"{len(idx)} strided rows, NumPy oracle"},
        "parity": {"rows": int(len(idx)), "max_rel_err": err, "tolerance": 1e-6},
    }), flush=True)


if __name__ == "__main__":
    main()
