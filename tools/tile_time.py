#!/usr/bin/env python3
"""Time of the observations-fastest LOO pass alone (C3 shape by default), best of a few calls, with a check of the result
against the draws-fastest pass.   OBS=1000000 DRAWS=4000 python tools/tile_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = int(os.environ.get("OBS", 1_000_000)), int(os.environ.get("DRAWS", 4000))
M = int(os.environ.get("TAIL", 190))
a = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(a, seed=3)
r0 = eng.psis_loo(a, M, "psis", 1.0, 0.7, pointwise=False)
ref = r0["agg"].clone()
view = a.t().contiguous().t()
del a
best = 1e9
eng.set_timing(True)
eng.kernel_ms()
for _ in range(int(os.environ.get("REPS", 6))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = eng.psis_loo(view, M, "psis", 1.0, 0.7, pointwise=False); torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
kms, kn = eng.kernel_ms()
rel = abs(ref[1].item() - r["agg"][1].item()) / abs(ref[1].item())
print(json.dumps({"obs_fastest_ms": round(best * 1e3, 3), "pass_ms_hip_events": round(kms / max(kn, 1), 3), "tb_per_s": round(N * S * 8 / best / 1e12, 3), "elpd_rel_diff": rel,
                  "slow_rows": r["agg"][7].item(), "kernels": eng.last_kernels()[:40], "lib": os.environ.get("PYLOO_AMD_LIB", "default")[-12:]}))
