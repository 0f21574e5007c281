import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyloo_amd.engine import get_engine
from pyloo_amd.base import tail_count_for
eng = get_engine(0)
for S, dt, n in ((20000, torch.float32, 40000), (8000, torch.float64, 40000)):
    t = torch.empty((n, S), dtype=dt, device="cuda")
    M = tail_count_for(S, 1.0)
    for off in (0.0, 0.05, 0.1, 0.3):
        for klo in (0.05, 0.2):
            eng.fill_synthetic_chains(t, seed=5, chains=4, rho=0.9, offset_sd=off, k_lo=klo, k_hi=0.5)
            for env in ("1", "0"):
                os.environ["PLA_NO_RETRY"] = env
            r = eng.psis_loo(t, M, "psis", 1.0, 0.7)
            torch.cuda.synchronize()
            a = r["agg"].cpu().numpy()
            print(f"S={S} off={off} k_lo={klo}: left to general kernel {int(a[7])} of {n} ({100*a[7]/n:.2f} %), high k {int(a[4])}")
