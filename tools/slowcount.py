#!/usr/bin/env python3
"""Rows the wave kernels hand to the general kernel, for a few synthetic populations
(PYLOO_AMD_LIB selects the build).   python tools/slowcount.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.base import tail_count_for
from pyloo_amd.engine import get_engine
eng = get_engine(0)
for name, N, S, dt, kw in (("C3 k in [0.05,0.6]", 1_000_000, 4000, torch.float64, dict(k_lo=0.05, k_hi=0.6)),
                           ("S=4000 k in [0.05,1.3]", 500_000, 4000, torch.float64, dict(k_lo=0.05, k_hi=1.3)),
                           ("S=4000 k in [1.0,1.3]", 200_000, 4000, torch.float64, dict(k_lo=1.0, k_hi=1.3)),
                           ("C5 shard", 125_000, 20000, torch.float32, dict(k_lo=0.05, k_hi=0.5, heavy_lo=1.0, heavy_hi=1.3)),
                           ("S=8000 f64", 100_000, 8000, torch.float64, dict(k_lo=0.05, k_hi=0.6))):
    t = torch.empty((N, S), dtype=dt, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0003, **kw)
    r = eng.psis_loo(t, tail_count_for(S, 1.0), "psis", 1.0, 0.7, pointwise=False)
    print(f"{name:28s} slow rows {int(r['agg'][7].item()):7d} of {N}  ({100.0 * r['agg'][7].item() / N:.3f} %)   k>0.7: {int(r['agg'][4].item())}")
    del t
    torch.cuda.empty_cache()
