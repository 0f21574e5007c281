"""Two passes over the same device matrix must agree bit for bit: prints how many observations differ (and the first few).
    python tools/determinism_check.py [--obs N] [--draws S] [--dtype f64|f32] [--layout draws|obs] [--repeat R] [--quiet]
(PYLOO_AMD_LIB / PLA_PIPE select what runs; --quiet prints only the passes that differ and a summary: soak runs)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyloo_amd.base import tail_count_for  # noqa: E402
from pyloo_amd.engine import get_engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--obs", type=int, default=1_000_000)
ap.add_argument("--draws", type=int, default=4000)
ap.add_argument("--repeat", type=int, default=4)
ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
ap.add_argument("--layout", default="draws", choices=["draws", "obs"])
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()
eng = get_engine(0)
t = torch.empty((a.obs, a.draws), dtype=torch.float64 if a.dtype == "f64" else torch.float32, device="cuda")
eng.fill_synthetic(t, seed=0x5EED0003)
if a.layout == "obs":
    t = t.t().contiguous().t()  # the same numbers, observations fastest in memory
n_bad = n_gave_up = 0
M = tail_count_for(a.draws, 1.0)
ref = eng.psis_loo(t, M, "psis", 1.0, 0.7)
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
for it in range(a.repeat):
    got = eng.psis_loo(t, M, "psis", 1.0, 0.7)
    torch.cuda.synchronize()
    line = []
    for k in ("diag", "loo_i", "lppd_i"):
        bad = torch.nonzero(~((ref[k] == got[k]) | (torch.isnan(ref[k]) & torch.isnan(got[k])))).flatten()
        line.append(f"{k}: {bad.numel()} differ")
        if bad.numel() and k == "diag":
            b = bad.cpu().numpy()
            import numpy as np
            runs = np.split(b, np.where(np.diff(b) != 1)[0] + 1)
            line.append(f"runs of consecutive rows: {len(runs)}, lengths {sorted(set(len(r) for r in runs))}, starts mod 16 {sorted(set(int(r[0]) % 16 for r in runs))}, "
                        f"chunk range {int(b.min()) // 16}..{int(b.max()) // 16}, zero in got {int((got[k][bad] == 0).sum())}, zero in ref {int((ref[k][bad] == 0).sum())}")
        if bad.numel():
            i = bad[:4].tolist()
            line.append(f"rows {i} ref {ref[k][bad[:4]].tolist()} got {got[k][bad[:4]].tolist()}")
    same_agg = torch.equal(ref["agg"], got["agg"])
    line.append("agg " + ("same" if same_agg else f"DIFFERS {ref['agg'].tolist()} {got['agg'].tolist()}"))
    gave_up = eng.stream_gave_up()
    differs = (not same_agg) or any(" 0 differ" not in x for x in line[:3] if "differ" in x)
    n_bad += int(differs)
    n_gave_up += int(gave_up)
    if differs or gave_up or not a.quiet:
        print(f"pass {it}: " + "; ".join(line), "| kernels:", eng.last_kernels()[:60], "| gave_up", gave_up, flush=True)
print(f"{a.repeat} passes over {a.obs} x {a.draws} {a.dtype} ({a.layout} fastest): {n_bad} differ from the first, {n_gave_up} streamed fits gave up; kernels: {eng.last_kernels()[:80]}")
