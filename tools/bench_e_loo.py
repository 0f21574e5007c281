#!/usr/bin/env python3
"""Side bench: e_loo on the device (pla_e_loo: weighted mean / variance + function-specific k; pla_e_loo_quantiles) on
device-resident matrices.   python tools/bench_e_loo.py   [OBS=200000]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = int(os.environ.get("OBS", 200_000)), 4000
ll = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(ll, seed=5)
lr = -ll
x = torch.randn((N, S), dtype=torch.float64, device="cuda")
lw, _ = eng.importance_weights(lr, 190, "psis")


def timed(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, r


t3, r = timed(lambda: eng.e_loo(x, lw, lr))
t2, _ = timed(lambda: eng.e_loo(x, lw))
tq, q = timed(lambda: eng.e_loo_quantiles(x, lw, np.array([0.05, 0.5, 0.95])))
print(json.dumps({"workload": f"f64 S={S} x N={N}, device-resident x / log-weights / log-ratios",
                  "e_loo_three_matrices_ms": t3, "tb_per_s": 3 * N * S * 8 / (t3 * 1e-3) / 1e12,
                  "e_loo_two_matrices_ms": t2, "tb_per_s_two": 2 * N * S * 8 / (t2 * 1e-3) / 1e12,
                  "quantiles_3_probs_ms": tq, "k_mean_unique": sorted(set(np.round(r["k_mean"].cpu().numpy(), 6).tolist()))[:3]}))
