#!/usr/bin/env python3
"""e_loo quantiles: time against the number of levels (what one level costs, what the set-up of a row costs).
python tools/quantile_levels_cost.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = 200_000, 4000
x = torch.empty((N, S), dtype=torch.float64, device="cuda")
lw = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(x, seed=5)
eng.fill_synthetic(lw, seed=6)
out = {}
for probs in ([0.5], [0.05, 0.5], [0.05, 0.5, 0.95], [0.05, 0.25, 0.5, 0.75, 0.95, 0.99]):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng.e_loo_quantiles(x, lw, np.array(probs)); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    out[len(probs)] = round(best * 1e3, 3)
print(json.dumps({"workload": f"f64 S={S} x N={N}", "ms_by_number_of_levels": out}))
