import sys, os
sys.path.insert(0, os.getcwd())
import torch
from pyloo_amd.engine import get_engine
eng = get_engine(0)
t = torch.empty((200000, 4000), dtype=torch.float64, device="cuda")
eng.fill_synthetic(t, seed=0x5EED0003)
r = eng.psis_loo(t, 190, "psis", 1.0, 0.7)
torch.cuda.synchronize()
print("slow rows:", r["agg"][7].item(), "of", t.shape[0])
