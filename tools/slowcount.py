#!/usr/bin/env python3
"""Rows the wave kernel hands to the general kernel on the bench matrix (PYLOO_AMD_LIB selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine
eng = get_engine(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = torch.empty((N, 4000), dtype=torch.float64, device="cuda")
eng.fill_synthetic(t, seed=0x5EED0003)
r = eng.psis_loo(t, 190, "psis", 1.0, 0.7, pointwise=False)
print(os.environ.get("PYLOO_AMD_LIB", "default"), "slow rows", int(r["agg"][7].item()), "of", N)
