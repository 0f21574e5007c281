#!/usr/bin/env python3
"""One LOO pass over a matrix with more than 2^32 elements (6 M x 4000 f64 = 192 GB of the 288 GB): 64-bit
indexing end to end, rows at the far end checked against the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import psis_oracle as orc
from pyloo_amd.engine import get_engine

eng = get_engine(0)
N, S = int(os.environ.get("BIG_N", 6_000_000)), 4000
t = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(t, seed=0x5EED0004)
torch.cuda.synchronize()
t0 = time.perf_counter()
res = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
idx = np.concatenate([np.arange(0, 64), np.arange(N // 2, N // 2 + 64), np.arange(N - 64, N)])
ref = orc.loo_pointwise(t[idx].cpu().numpy(), 1.0)
err = {k2: float(np.max(np.abs(res[k1][idx].cpu().numpy() - ref[k2]) / np.maximum(np.abs(ref[k2]), 1e-2)))
       for k1, k2 in (("diag", "diag"), ("loo_i", "loo_i"), ("lppd_i", "lppd_i"))}
agg = res["agg"].cpu().numpy()
print(f"N={N} ({N * S * 8 / 1e9:.0f} GB): {dt * 1e3:.1f} ms, {N / dt / 1e6:.1f} M obs/s, n={int(agg[0])}, slow={int(agg[7])}, "
      f"all finite={bool(torch.isfinite(res['loo_i']).all())}, max rel err at both ends and the middle {err}")
assert int(agg[0]) == N and max(err.values()) < 1e-8
