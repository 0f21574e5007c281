#!/usr/bin/env python3
"""Cost of the ArviZ-native layout (observations fastest) on a device-resident matrix: the
lane-per-observation LOO kernels (pla_col.h, the matrix read in place), the tiled transposing ingestion they replace
(PLA_INGEST_TRANSPOSE=1 in the environment: round 1's path; WAIC still uses it) and torch's copy, next to the draws-fastest
pass.   python tools/obs_fastest_cost.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = int(os.environ.get("OBS", 1_000_000)), 4000
a = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(a, seed=3)
b = a.t().contiguous()   # (S, N): observations fastest
view = b.t()             # (N, S) view with strides (1, N)


def timed(f, n=4):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, r


t_rm, r0 = timed(lambda: eng.psis_loo(a, 190, "psis", 1.0, 0.7, pointwise=False))
t_of, r1 = timed(lambda: eng.psis_loo(view, 190, "psis", 1.0, 0.7, pointwise=False))
t_wa, _ = timed(lambda: eng.waic(view, 1.0, pointwise=False))
del a
t_torch, _ = timed(lambda: view.contiguous(), 2)
eng.psis_loo(view, 190, "psis", 1.0, 0.7, pointwise=False)
path = "transposing ingestion" if os.environ.get("PLA_INGEST_TRANSPOSE", "0") not in ("", "0") else eng.last_kernels()
rel = abs(r0["agg"][1].item() - r1["agg"][1].item()) / abs(r0["agg"][1].item())
print(json.dumps({"workload": f"f64 S={S} x N={N}, device-resident", "obs_fastest_loo_path": path, "loo_draws_fastest_ms": t_rm,
                  "loo_obs_fastest_ms": t_of, "obs_fastest_tb_per_s_algorithmic": N * (S * 8 + 24) / (t_of * 1e-3) / 1e12,
                  "waic_obs_fastest_ms": t_wa, "torch_contiguous_copy_ms": t_torch, "extra_over_draws_fastest_ms": t_of - t_rm,
                  "elpd_rel_diff": rel, "rows_to_general_kernel": r1["agg"][7].item()}))
