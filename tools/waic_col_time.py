#!/usr/bin/env python3
"""WAIC on an observations-fastest device matrix (C3 shape), best of a few calls.  python tools/waic_col_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = int(os.environ.get("OBS", 1_000_000)), int(os.environ.get("DRAWS", 4000))
b = torch.empty((S, N), dtype=torch.float64, device="cuda")
eng.fill_synthetic(b.view(-1)[: (S * N // 4000) * 4000].view(-1, 4000), seed=3)
view = b.t()
best = 1e9
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = eng.waic(view, 1.0, pointwise=False); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
print(json.dumps({"waic_obs_fastest_ms": round(best * 1e3, 3), "tb_per_s": round(N * S * 8 / best / 1e12, 3), "lib": os.environ.get("PYLOO_AMD_LIB", "default")[-12:]}))
