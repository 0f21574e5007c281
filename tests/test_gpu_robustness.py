"""Rows that are NOT exchangeable along the draw axis (``-m gpu``).

``pl.loo(idata)`` stacks ``(chain, draw)`` chain-major (loo.py:189): real rows are autocorrelated MCMC output, chains can
differ in location or scale, and a user may hand over sorted draws.  The wave kernels take their speculative candidate
threshold from a SAMPLE of the row (pla_fast.h: bitrev_order); these tests hold the results to the oracle at the usual
tolerance whatever the order of the draws, and bound the fraction of rows the fast selection path hands to the general
kernel (``agg[7]``), which is what a biased sample would cost."""

import json
import os
import zlib

import numpy as np
import pytest

from oracle import psis_oracle as orc
from test_gpu_parity import close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from pyloo_amd.engine import get_engine

    return get_engine(0)


def _ar1(rng, n, s, rho):
    z = np.empty((n, s))
    z[:, 0] = rng.normal(size=n)
    e = rng.normal(size=(n, s)) * np.sqrt(1.0 - rho * rho)
    for t in range(1, s):
        z[:, t] = rho * z[:, t - 1] + e[:, t]
    return z


def _exp_from_normal(z):
    """Exp(1) marginals with the dependence of z: E = -log(1 - Phi(z))."""
    from scipy.special import log_ndtr

    return -log_ndtr(-z)


def make_rows(kind, n, s, rng, chains=4):
    k = rng.uniform(0.1, 0.9, size=n)[:, None]
    c = rng.normal(size=(n, 1))
    if kind == "iid":
        e = rng.exponential(size=(n, s))
    elif kind in ("ascending", "descending"):
        e = np.sort(rng.exponential(size=(n, s)), axis=1)
        if kind == "descending":
            e = e[:, ::-1]
    elif kind == "ar1":  # one long autocorrelated chain per row
        e = _exp_from_normal(_ar1(rng, n, s, 0.9))
    elif kind == "chains_ar1":  # chain-major stack of autocorrelated chains with their own offsets
        per = s // chains
        e = np.concatenate([_exp_from_normal(_ar1(rng, n, per, 0.9)) for _ in range(chains)], axis=1)
        e = e + np.repeat(rng.normal(scale=0.3, size=(n, chains)), per, axis=1) / k  # shifts the chain's log-likelihoods
    elif kind == "chains_scale":  # chains of different scale: one chain owns most of the tail
        per = s // chains
        scale = np.repeat(np.array([1.0, 1.3, 0.8, 1.1])[None, :chains], per, axis=1)
        e = rng.exponential(size=(n, s)) * scale
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(-k * e + c)


# (kind, S, dtype, bound on the fraction of rows handed to the general kernel)
# Measured (profiles/r02_handover_rates_*.jsonl).  Round 1's threshold -- group maxima over the row's first 512 draws -- handed
# over 100 % of sorted rows, 13 % of AR(1) rows and 40 % of chain-major AR(1) rows with per-chain offsets.  Now the sample is
# spread over the row (bit-reversed vector order) and the threshold is verified against an EXACT count of the register
# block before the sweep, with bisection on such counts when it fails: every one-chunk case below stays on the fast path.
CASES = [
    ("iid", 4000, np.float64, 0.002),
    ("ascending", 4000, np.float64, 0.002),
    ("descending", 4000, np.float64, 0.002),
    ("ar1", 4000, np.float64, 0.002),
    ("chains_ar1", 4000, np.float64, 0.002),
    ("chains_scale", 4000, np.float64, 0.002),
    ("chains_ar1", 2000, np.float64, 0.002),
    # long rows are read ONCE in chunks of 4096 draws, so the threshold can only know the first chunk: exact for that
    # chunk (autocorrelation inside it no longer matters: 79 % -> 3 %), blind to what later chains do differently
    # ... and a row that ends with too few / too many draws above it comes round a second time with a threshold corrected by
    # the first attempt's exact counts (pla_chunked.h, ChunkRetry): 4.5 % -> 0.2 % and 15.8 % -> 1.9 % (measured, round 3)
    ("ar1", 20000, np.float32, 0.01),
    ("chains_ar1", 8000, np.float64, 0.03),
]


@pytest.mark.parametrize("kind,S,dt,bound", CASES)
def test_order_of_the_draws(eng, kind, S, dt, bound):
    n = 3000 if S <= 4000 else 1200
    rng = np.random.default_rng(zlib.crc32(f"{kind}{S}".encode()))
    ll = make_rows(kind, n, S, rng).astype(dt)
    M = orc.tail_count(S, 1.0)
    res = eng.psis_loo(ll, M, "psis", 1.0, 0.7)
    frac = res["agg"][7] / n
    if os.environ.get("PLA_HANDOVER_LOG"):  # (a record for profiles/, only when asked for: no side effects otherwise)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", os.environ["PLA_HANDOVER_LOG"]), "a") as f:
            f.write(json.dumps({"kind": kind, "S": S, "dtype": np.dtype(dt).name, "rows": n, "handed_over": float(res["agg"][7]),
                                "fraction": float(frac)}) + "\n")
    idx = np.arange(0, n, 15)
    ref = orc.loo_arrays(ll[idx].astype(np.float64), 1.0)
    close(res["diag"][idx], ref["khat"], what="khat")
    close(res["loo_i"][idx], ref["loo_i"], what="loo_i")
    close(res["lppd_i"][idx], ref["lppd_i"], what="lppd_i")
    assert frac <= bound, f"{kind} S={S}: {res['agg'][7]:.0f} of {n} rows left the fast selection path"


@pytest.mark.parametrize("kind,S", [("iid", 4000), ("ascending", 4000), ("descending", 4000), ("ar1", 4000), ("chains_ar1", 4000),
                                    ("chains_scale", 4000), ("chains_ar1", 2000)])
def test_order_of_the_draws_observations_fastest(eng, kind, S):
    """The same orders of draws on a device matrix with the observations fastest (pla_tile.h): its threshold comes from 512 draws
    in 128 clusters spread over the row, and an observation whose list comes out too short or too long is swept a second time
    with a threshold corrected from the exact count -- the general kernel (1 us per observation in this layout) sees next to nothing."""
    import torch

    n = 3000
    rng = np.random.default_rng(zlib.crc32(f"{kind}{S}".encode()))
    ll = make_rows(kind, n, S, rng)
    M = orc.tail_count(S, 1.0)
    view = torch.from_numpy(np.ascontiguousarray(ll.T)).cuda().T
    assert view.stride(0) == 1
    res = eng.psis_loo(view, M, "psis", 1.0, 0.7)
    assert "tile_loo_kernel" in eng.last_kernels()
    frac = res["agg"][7].item() / n
    idx = np.arange(0, n, 15)
    ref = orc.loo_arrays(ll[idx], 1.0)
    close(res["diag"].cpu().numpy()[idx], ref["khat"], what="khat")
    close(res["loo_i"].cpu().numpy()[idx], ref["loo_i"], what="loo_i")
    close(res["lppd_i"].cpu().numpy()[idx], ref["lppd_i"], what="lppd_i")
    assert frac <= 0.004, f"{kind} S={S}: {res['agg'][7].item():.0f} of {n} rows left the fast selection path"


def test_chain_major_weights(eng):
    """Same for the weights flavour (psislw): chain-major autocorrelated rows, smoothed log-weights against the oracle."""
    rng = np.random.default_rng(5)
    ll = make_rows("chains_ar1", 400, 4000, rng)
    lw, k = eng.importance_weights(-ll, 190, "psis")
    want_lw, want_k = orc.importance_weights(-ll[::8], "psis", 1.0)
    close(k[::8], want_k, what="khat")
    close(lw[::8], want_lw, what="lw")
