#!/usr/bin/env python3
"""Why rows leave the fast selection path, per kind of row order (profiling build with PLA_WAVE_ABLATE=1):
   python -m pyloo_amd.build --alt=ablate -DPLA_WAVE_ABLATE=1 -DPLA_EXPERIMENT && PYLOO_AMD_LIB=$PWD/pyloo_amd/lib/alt_ablate.so PLA_PRINT_REASONS=1 python tools/slow_reasons.py"""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from test_gpu_robustness import CASES, make_rows  # noqa: E402

from oracle import psis_oracle as orc  # noqa: E402
from pyloo_amd.engine import get_engine  # noqa: E402

eng = get_engine(0)
for kind, S, dt, bound in CASES:
    n = 3000 if S <= 4000 else 1200
    rng = np.random.default_rng(zlib.crc32(f"{kind}{S}".encode()))
    ll = torch.from_numpy(make_rows(kind, n, S, rng).astype(dt)).cuda()
    print(f"== {kind} S={S} {np.dtype(dt).name}", flush=True)
    res = eng.psis_loo(ll, orc.tail_count(S, 1.0), "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    print(f"   handed over: {int(res['agg'][7].item())} of {n}", flush=True)
