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

# the same matrix in ArviZ's native order (chain, draw, obs): pl.loo() hands the engine an observations-fastest VIEW; the
# alternative is the transposing copy on the host that stack_samples() used to make
import warnings
import pyloo_amd as pl
native = np.ascontiguousarray(host.T).reshape(4, S // 4, N)
del host
warnings.simplefilter("ignore")
pl.loo(native[:, :, :1000], reff=1.0)
t0 = time.perf_counter(); out = pl.loo(native, reff=1.0); t1 = time.perf_counter()
t2 = time.perf_counter(); copy = np.ascontiguousarray(native.reshape(S, N).T); t3 = time.perf_counter()
t4 = time.perf_counter(); r2 = eng.psis_loo(copy, 190, "psis", 1.0, 0.7); t5 = time.perf_counter()
print("pl.loo on (chain, draw, obs) host array, %.1f GB: %.3f s end to end (%.1f GB/s); the host transpose alone takes %.3f s "
      "(+ %.3f s for the pass on the copy); same elpd: %s" % (native.nbytes / 1e9, t1 - t0, native.nbytes / 1e9 / (t1 - t0), t3 - t2,
                                                              t5 - t4, bool(abs(out["elpd_loo"] - r2["agg"][1]) < 1e-9 * abs(r2["agg"][1]))))
