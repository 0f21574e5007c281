#!/usr/bin/env python3
"""PCIe-inclusive rate of the PLA_HOST path (NumPy array in host memory -> pla_psis_loo): for DESIGN.md only."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine
eng = get_engine(0)
N, S = 100_000, 4000
t = torch.empty((N, S), dtype=torch.float64, device="cuda")
eng.fill_synthetic(t, seed=0x5EED0002)
host = t.cpu().numpy()
del t
eng.psis_loo(host[:1000], 190, "psis", 1.0, 0.7)
t0 = time.perf_counter(); r = eng.psis_loo(host, 190, "psis", 1.0, 0.7); t1 = time.perf_counter()
print("PLA_HOST: %d x %d f64 (%.1f GB) in %.3f s = %.1f GB/s, %.0f obs/s" % (N, S, host.nbytes/1e9, t1-t0, host.nbytes/1e9/(t1-t0), N/(t1-t0)))
