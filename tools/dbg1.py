import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from pyloo_amd.engine import get_engine
eng = get_engine(0)
g = np.load('tests/golden/s2000_r1_f32.npz')
ll = g['ll'][8:9]
for name, arr in (("f32", ll), ("f64-upcast", ll.astype(np.float64))):
    r = eng.psis_loo(np.repeat(arr, 4, axis=0), 135, "psis", 1.0, 0.7)
    print(name, repr(r["diag"][0]), r["loo_i"][0], r["agg"][7], "want", g["khat"][8], g["loo_i"][8])
