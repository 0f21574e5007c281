"""Seeded random sweep over shapes, dtypes, tail counts and row families (``-m gpu``): every entry point of the hot path against
the oracle on the same inputs.  The fixed cases elsewhere pin known edges; this one walks the dispatch table -- one-chunk,
chunked, column, general kernels; LOO, weights, SIS / TIS, WAIC, e_loo -- with shapes nobody picked by hand."""

import os

import numpy as np
import pytest

from oracle import psis_oracle as orc
from test_gpu_parity import close, has_tail_ties

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pyloo_amd.engine import get_engine

    return get_engine(0)


def make_rows(rng, n, s, dt):
    """Log-likelihood rows of mixed character: light / heavy tails, autocorrelated, sorted, tied, shifted chains, outliers."""
    k = rng.uniform(0.05, 1.2, size=(n, 1))
    ll = -k * rng.exponential(size=(n, s)) + rng.normal(size=(n, 1)) * 3.0
    for i in range(n):
        kind = rng.integers(0, 8)
        if kind == 1:                                   # AR(1) in the draws
            e = rng.normal(size=s)
            z = np.empty(s)
            z[0] = e[0]
            for j in range(1, s):
                z[j] = 0.85 * z[j - 1] + e[j]
            ll[i] = -0.5 * (z * 0.6) ** 2
        elif kind == 2:
            ll[i] = np.sort(ll[i])[:: rng.choice([-1, 1])]
        elif kind == 3:
            ll[i] = np.round(ll[i] * 8.0) / 8.0         # ties
        elif kind == 4:                                 # chain-major: four chains, shifted and scaled
            q = s // 4
            for c in range(4):
                ll[i, c * q:(c + 1) * q] = ll[i, c * q:(c + 1) * q] * rng.uniform(0.6, 1.6) + rng.normal() * 0.5
        elif kind == 5:
            ll[i, rng.integers(0, s)] -= rng.uniform(5.0, 60.0)   # one draw far in the tail of the ratios
        elif kind == 6:
            ll[i] *= rng.uniform(10.0, 200.0)           # wide range (some beyond 690 nats)
    return ll.astype(dt)


CONFIGS = [(100 * rep + seed, s, n, reff, dt) for rep in range(int(os.environ.get("PYLOO_AMD_FUZZ_REPS", "3"))) for seed, (s, n, reff, dt) in enumerate([
    (256, 37, 1.0, np.float64), (258, 5, 0.7, np.float64), (640, 64, 1.3, np.float32), (1000, 33, 0.5, np.float64),
    (1536, 20, 1.0, np.float32), (2000, 48, 0.31, np.float64), (3000, 17, 1.0, np.float64), (4000, 70, 0.9, np.float32),
    (4096, 9, 1.0, np.float64), (4098, 12, 1.0, np.float64), (4352, 10, 0.8, np.float32), (5000, 21, 1.0, np.float64),
    (6144, 8, 0.4, np.float64), (8000, 30, 1.0, np.float32), (8192, 6, 1.0, np.float64), (10000, 14, 0.6, np.float64),
    (12000, 10, 1.0, np.float32), (16384, 5, 1.0, np.float64), (20000, 12, 1.0, np.float32), (20000, 7, 0.25, np.float64),
    (24000, 4, 1.0, np.float64), (66000, 3, 1.0, np.float32)])]


@pytest.mark.parametrize("seed,S,N,reff,dt", CONFIGS)
def test_random_rows_every_entry_point(eng, seed, S, N, reff, dt):
    rng = np.random.default_rng(1000 + seed)
    ll = make_rows(rng, N, S, dt)
    ll64 = ll.astype(np.float64)
    M = orc.tail_count(S, reff)
    with np.errstate(all="ignore"):
        ref = orc.loo_arrays(ll64, reff)
    # ---- LOO pass (host array, device tensor, observations-fastest device view) ----
    res = eng.psis_loo(ll, M, "psis", 1.0, ref["good_k"])
    for key, want in (("diag", "khat"), ("loo_i", "loo_i"), ("lppd_i", "lppd_i")):
        close(res[key], ref[want], what=f"{key} S={S}")
    import torch

    t = torch.from_numpy(ll).cuda()
    dev = eng.psis_loo(t, M, "psis", 1.0, ref["good_k"])
    close(dev["loo_i"].cpu().numpy(), ref["loo_i"], what="loo_i (device)")
    view = t.t().contiguous().t()  # (N, S) view of an (S, N) buffer
    col = eng.psis_loo(view, M, "psis", 1.0, ref["good_k"])
    close(col["loo_i"].cpu().numpy(), ref["loo_i"], what="loo_i (observations fastest)")
    close(col["diag"].cpu().numpy(), ref["khat"], what="khat (observations fastest)")
    # ---- weights ----
    lw, kk = eng.importance_weights(-ll, M, "psis")
    close(kk, ref["khat"], what="khat (weights)")
    want = ref["lw"].copy()
    got = np.asarray(lw, dtype=np.float64).copy()
    for i in range(N):
        if has_tail_ties(ll[i], M):
            got[i], want[i] = np.sort(got[i]), np.sort(want[i])
    if dt == np.float64:
        close(got, want, what="lw")
    else:
        close(got.astype(np.float32), want.astype(np.float32), rtol=3e-7, atol=2e-7, what="lw (f32)")
    # ---- SIS / TIS, WAIC ----
    for method in ("sis", "tis"):
        with np.errstate(all="ignore"):
            r2 = orc.loo_pointwise(ll64, 1.0, method)
        g2 = eng.psis_loo(ll, 0, method, 1.0, 0.7)
        close(g2["loo_i"], r2["loo_i"], what=f"loo_i ({method})")
        close(g2["diag"], r2["diag"], rtol=1e-8, what=f"ess ({method})")
        lw2, ess2 = eng.importance_weights(-ll, 0, method)   # the weights-returning flavour of the same kernel
        close(ess2, r2["diag"], rtol=1e-8, what=f"ess (weights, {method})")
        if dt == np.float64:
            close(lw2, r2["lw"], what=f"lw ({method})")
        else:
            close(lw2, r2["lw"].astype(np.float32), rtol=3e-7, atol=2e-7, what=f"lw ({method}, f32)")
    with np.errstate(all="ignore"):
        w = orc.waic_arrays(ll64)
    gw = eng.waic(ll)
    close(gw["waic_i"], w["waic_i"], rtol=1e-8, what="waic_i")
    # ---- e_loo on the smoothed weights ----
    x = (rng.normal(size=(N, S)) * 2.0 + 0.3).astype(dt)
    ok = np.isfinite(np.asarray(lw, dtype=np.float64)).all(axis=1)
    if ok.any():
        with np.errstate(all="ignore"):
            e = orc.e_loo_arrays(x[ok].astype(np.float64), np.asarray(lw, dtype=np.float64)[ok], -ll64[ok])
        ge = eng.e_loo(x[ok], np.ascontiguousarray(lw[ok]), np.ascontiguousarray(-ll[ok]))
        close(ge["mean"], e["mean"], rtol=1e-8, atol=1e-9, what="e_loo mean")
        close(ge["k_mean"], e["k_mean"], rtol=1e-12, what="e_loo k")
