"""BASELINE.json configurations at their full per-GPU sizes (``-m gpu``), device-generated with the SURVEY section 8(d)
generator: C3 (1 M x 4000 f64, the configuration the metric is quoted on; also the per-GPU shard of C4) and one C5 shard
(125 000 x 20 000 f32, 30 % heavy-tailed rows).  The oracle finishes a strided sample in seconds; the whole matrix is
covered by size-independent properties (bitwise determinism, the reductions against ``math.fsum``, loo_i <= lppd_i).

Also: the multi-rank flow of ``bench.py --gpus N`` through its own launcher (two ranks on one card over gloo -- RCCL
refuses two ranks on one device) and the RCCL collective itself at world size 1."""

import json
import math
import os
import subprocess
import sys

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


def _check_whole_matrix(eng, t, M, good_k, n_sample, slow_bound, heavy_fraction=None):
    import torch

    N, S = t.shape
    a = eng.psis_loo(t, M, "psis", 1.0, good_k)
    b = eng.psis_loo(t, M, "psis", 1.0, good_k)
    torch.cuda.synchronize()
    for key in ("diag", "loo_i", "lppd_i", "agg"):
        assert torch.equal(a[key], b[key]), key  # bitwise reproducible
    del b
    loo_i, lppd_i, diag = (a[k].cpu().numpy() for k in ("loo_i", "lppd_i", "diag"))
    agg = a["agg"].cpu().numpy()
    assert np.all(np.isfinite(loo_i)) and np.all(np.isfinite(lppd_i)) and np.all(np.isfinite(diag))
    assert np.all(loo_i <= lppd_i + 1e-9)
    np.testing.assert_allclose(agg[1], math.fsum(loo_i), rtol=1e-12)
    np.testing.assert_allclose(agg[3], math.fsum(lppd_i), rtol=1e-12)
    np.testing.assert_allclose(agg[2] / N, np.var(loo_i), rtol=1e-9)
    assert agg[0] == N and agg[4] == np.sum(diag > good_k)
    assert agg[7] <= slow_bound * N, (agg[7], N)  # rows that left the fast selection path
    if heavy_fraction is not None:
        assert abs(agg[4] / N - heavy_fraction) < 0.03, agg[4] / N
    idx = np.arange(0, N, max(1, N // n_sample))[:n_sample]
    rows = t[torch.from_numpy(idx).to(t.device)].cpu().numpy().astype(np.float64)  # f32 input: parity vs the upcast data
    ref = orc.loo_arrays(rows, 1.0)
    close(diag[idx], ref["khat"], what="khat sample")
    close(loo_i[idx], ref["loo_i"], what="loo_i sample")
    close(lppd_i[idx], ref["lppd_i"], what="lppd_i sample")
    return agg


def test_c3_full_size(eng):
    """C3 = C4's per-GPU shard: S=4000 x N=1 000 000 f64 (32 GB), seed 0x5EED0003, 1 000 rows against the oracle."""
    import torch

    S, N = 4000, 1_000_000
    t = torch.empty((N, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0003)
    agg = _check_whole_matrix(eng, t, orc.tail_count(S, 1.0), 0.7, 1000, 0.0005)
    assert agg[5] == 0
    del t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n_obs", [5, 1000, 200_003])
def test_streamed_pass_equals_the_kernels_back_to_back(eng, n_obs, monkeypatch):
    """The streamed split pass (fit kernel beside the wave kernel, chunks handed over behind per-chunk flags inside one
    launch) against the same two kernels run one after the other (PLA_PIPE=0): every output of every observation bit for bit,
    several times over -- a stale read of the hand-over would show as a differing row.  Includes rows for the general kernel
    and a row count that is not a multiple of the chunk."""
    import torch

    S = 4000
    t = torch.empty((n_obs, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0002, k_lo=0.05, k_hi=0.9)
    t[min(7, n_obs - 1), 11] = float("inf")
    t[n_obs // 2, 5] = -float("inf")
    # a second matrix, visited in turn with the first: the hand-over buffer is reused by every pass, so a stale read would
    # deliver the OTHER matrix's tail -- which the same matrix twice in a row could never show
    other = torch.empty_like(t)
    eng.fill_synthetic(other, seed=0x5EED0009, k_lo=0.05, k_hi=0.5)
    M = orc.tail_count(S, 1.0)

    def run(pipe, m=None):
        monkeypatch.setenv("PLA_PIPE", pipe)
        r = eng.psis_loo(t if m is None else m, M, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
        return {k: r[k].clone() for k in ("diag", "loo_i", "lppd_i", "agg")}

    ref, ref_other = run("0"), run("0", other)
    assert not torch.equal(ref["loo_i"], ref_other["loo_i"])
    for _ in range(4):
        got_other = run("1", other)
        got = run("1")
        for k in ref:
            same = (ref[k] == got[k]) | (torch.isnan(ref[k]) & torch.isnan(got[k]))
            assert bool(same.all()), (k, int((~same).sum()))
            assert torch.equal(ref_other[k], got_other[k]), ("other matrix", k)
    del other
    assert eng.stream_gave_up() == 0  # (no streamed pass so far had to fall back)
    # the fit kernel gives up waiting after ONE microsecond without its chunk (what a profiler that serialises the two kernels
    # the wrong way round causes, after its 2 / 20 ms): whatever it left is fitted by the plain fit kernel behind it -- same
    # bits -- and the engine's statistics say that it happened
    monkeypatch.setenv("PLA_STREAM_PATIENCE_US", "1")
    got = run("1")
    monkeypatch.delenv("PLA_STREAM_PATIENCE_US")
    for k in ref:
        same = (ref[k] == got[k]) | (torch.isnan(ref[k]) & torch.isnan(got[k]))
        assert bool(same.all()), ("gave up", k, int((~same).sum()))
    gave_up = eng.stream_gave_up()
    if n_obs >= 1000:  # (with five rows the sweep may be over before the fit kernel looks at all)
        assert gave_up >= 1
    assert eng.stream_gave_up() == 0  # (read and reset)
    # and against the oracle (a sample)
    idx = np.unique(np.linspace(0, n_obs - 1, 40).astype(np.int64))
    rows = t[torch.from_numpy(idx).to(t.device)].cpu().numpy()
    want = orc.loo_arrays(rows, 1.0)
    close(got["diag"].cpu().numpy()[idx], want["khat"], what="khat")
    close(got["loo_i"].cpu().numpy()[idx], want["loo_i"], what="loo_i")
    del t
    torch.cuda.empty_cache()


def test_calls_on_two_streams_of_one_engine_do_not_share_the_workspace(eng):
    """re-entrancy per (device, stream).  One engine = one workspace (hand-over buffer, flags, row
    lists), so the engine orders its calls across streams itself: every call records an event behind its work and a call on
    another stream waits for it.  Two matrices, passes enqueued alternately on two streams with nothing in between on the
    host: every result equals the one-stream result bit for bit (before: whichever pass came second wrote into the hand-over
    the first was still reading)."""
    import torch

    S, n = 4000, 150_001
    M = orc.tail_count(S, 1.0)
    mats = []
    for seed in (0x5EED0011, 0x5EED0012):
        t = torch.empty((n, S), dtype=torch.float64, device="cuda")
        eng.fill_synthetic(t, seed=seed, k_lo=0.05, k_hi=0.8)
        mats.append(t)
    torch.cuda.synchronize()
    ref = []
    for t in mats:
        r = eng.psis_loo(t, M, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
        ref.append({k: r[k].clone() for k in ("diag", "loo_i", "lppd_i", "agg")})
    assert not torch.equal(ref[0]["loo_i"], ref[1]["loo_i"])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = []
    for rep in range(3):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                got.append((i, eng.psis_loo(mats[i], M, "psis", 1.0, 0.7)))
            with torch.cuda.stream(streams[1 - i]):  # and the other layout's kernels of the same workspace on the other stream
                got.append((1 - i, eng.psis_loo(mats[1 - i], M, "psis", 1.0, 0.7)))
    torch.cuda.synchronize()
    for i, r in got:
        for k in ref[i]:
            assert torch.equal(ref[i][k], r[k]), (i, k)
    del mats, got, ref
    torch.cuda.empty_cache()


def test_streamed_pass_full_block_at_the_widest_hand_over(eng, monkeypatch):
    """A full block of 2^20 observations at the widest hand-over stride of the streamed pass (tail counts 193..250 -> 256
    doubles per observation: reff = 0.7 at S = 4000): the block's hand-over ends at 2^31 bytes, the edge of what the fit
    kernel's 32-bit buffer offsets reach.  Every row of the block (and the rows of the short block behind it) against the two
    kernels back to back."""
    import torch

    S, n = 4000, (1 << 20) + 77
    M = orc.tail_count(S, 0.7)
    assert 192 < M <= 250
    t = torch.empty((n, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED000A)
    out = {}
    for pipe in ("0", "1"):
        monkeypatch.setenv("PLA_PIPE", pipe)
        r = eng.psis_loo(t, M, "psis", 1.0, 0.7)
        torch.cuda.synchronize()
        out[pipe] = {k: r[k].clone() for k in ("diag", "loo_i", "lppd_i", "agg")}
    for k in out["0"]:
        assert torch.equal(out["0"][k], out["1"][k]), k
    idx = np.array([0, 1, (1 << 20) - 2, (1 << 20) - 1, 1 << 20, n - 1])
    ref = orc.loo_arrays(t[torch.from_numpy(idx).cuda()].cpu().numpy(), 0.7)
    close(out["1"]["diag"].cpu().numpy()[idx], ref["khat"], what="khat")
    close(out["1"]["loo_i"].cpu().numpy()[idx], ref["loo_i"], what="loo_i")
    del t
    torch.cuda.empty_cache()


def test_observations_fastest_more_than_one_block(eng):
    """The tile kernel of the observations-fastest path (pla_tile.h) runs in blocks of 2^20 observations, streamed: flags, work
    counter and hand-over buffer are reused from block to block, the second block starts in the middle of the matrix's rows.
    2^20 + 16 * 37 + 5 observations (a ragged last group) against the draws-fastest pass, every observation."""
    import torch

    S, n = 4000, (1 << 20) + 16 * 37 + 5
    M = orc.tail_count(S, 1.0)
    t = torch.empty((n, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED000B)
    b = eng.psis_loo(t, M, "psis", 1.0, 0.7)
    ref = {k: b[k].clone() for k in ("diag", "loo_i", "lppd_i", "agg")}
    view = t.T.contiguous().T
    del t
    torch.cuda.empty_cache()
    a = eng.psis_loo(view, M, "psis", 1.0, 0.7)
    assert "tile_loo_kernel<SYNC>" in eng.last_kernels()
    torch.cuda.synchronize()
    for k in ("diag", "loo_i", "lppd_i"):
        assert torch.allclose(a[k], ref[k], rtol=1e-10, atol=1e-11), k
    assert abs(a["agg"][1].item() - ref["agg"][1].item()) <= 1e-10 * abs(ref["agg"][1].item())
    assert a["agg"][7].item() <= 1e-4 * n
    del view
    torch.cuda.empty_cache()


def test_c5_shard(eng):
    """One GPU's shard of C5: S=20 000 x N=125 000 f32 (10 GB), seed 0x5EED0005, rows with i mod 10 in {0, 3, 6} drawn
    with k in [1, 1.3): ~30 % of the observations end above khat = 0.7.  500 rows against the oracle on the upcast data."""
    import torch

    S, N = 20000, 125_000
    t = torch.empty((N, S), dtype=torch.float32, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0005, k_lo=0.05, k_hi=0.5, heavy_lo=1.0, heavy_hi=1.3)
    M = orc.tail_count(S, 1.0)
    assert M == 425
    # heavy rows whose raw tail cancels against the sum of all exponentials go to the general kernel
    _check_whole_matrix(eng, t, M, 0.7, 500, 0.005, heavy_fraction=0.30)
    del t
    torch.cuda.empty_cache()


def _run_bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_two_ranks(eng):
    """`python bench.py --gpus 2` (no launcher around it) brings up two ranks itself, shards the observations, merges the
    aggregates with the one all-reduce and prints ONE line from rank 0.  On a one-GPU box both ranks share the card and
    reduce over gloo; the driver's multi-GPU node runs the same code over RCCL."""
    import torch

    n_local, S = 20000, 4000
    out = _run_bench(["--gpus", "2", "--obs", str(n_local), "--steps", "2", "--warmup", "1", "--no-cpu"],
                     {"PYLOO_AMD_BENCH_BACKEND": "gloo", "PYLOO_AMD_BENCH_DEVICE": "0"})
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["backend"] == "gloo" and out["scaling"] == "weak"
    assert out["devices_seen"] == [0, 0] and out["kernel_ms_per_rank"]["min"] <= out["kernel_ms_per_rank"]["max"]
    assert "wave_loo_kernel" in out["roofline"]["kernels"]
    assert out["value"] > 0 and abs(out["value"] - 2 * n_local * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    t = torch.empty((2 * n_local, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0003)  # rank r generated rows [r n_local, (r + 1) n_local) of this matrix
    agg = eng.psis_loo(t, 190, "psis", 1.0, 0.7, pointwise=False)["agg"].cpu().numpy()
    np.testing.assert_allclose(out["config"]["elpd_loo"], agg[1], rtol=1e-12)
    assert out["config"]["n_high_k"] == agg[4]


def test_bench_launches_four_ranks(eng):
    """The flow the driver runs at 2, 4 and 8 GPUs, rehearsed at four ranks on one card over gloo (this pool allows six
    processes on a card, the test runner being one of them: the 8-rank case is the driver's to start): rank count, per-rank
    records, the label (a reduced rehearsal is not C4: tests/test_sharded_gloo.py checks the C4 switch through
    bench.resolve_workload), the merged aggregates against one pass over the whole matrix."""
    import torch

    n_local, S = 4000, 4000
    out = _run_bench(["--gpus", "4", "--obs", str(n_local), "--steps", "2", "--warmup", "1", "--no-cpu"],
                     {"PYLOO_AMD_BENCH_BACKEND": "gloo", "PYLOO_AMD_BENCH_DEVICE": "0"})
    assert out["n_gpus"] == 4 and out["ranks"] == 4 and out["backend"] == "gloo" and out["scaling"] == "weak"
    assert out["devices_seen"] == [0, 0, 0, 0] and out["kernel_ms_per_rank"]["min"] <= out["kernel_ms_per_rank"]["max"]
    assert out["config"]["workload"].startswith("custom:") and out["config"]["seed"] == "0x5eed0003"
    assert {"PYLOO_AMD_BENCH_BACKEND=gloo", "PYLOO_AMD_BENCH_DEVICE=0"} <= set(out["env_overrides"]) and out["library_env_overrides"] == ""
    assert out["stream_gave_up"] == 0
    assert abs(out["value"] - 4 * n_local * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    t = torch.empty((4 * n_local, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0003)  # rank r generated rows [r n_local, (r + 1) n_local) of this matrix
    agg = eng.psis_loo(t, 190, "psis", 1.0, 0.7, pointwise=False)["agg"].cpu().numpy()
    np.testing.assert_allclose(out["config"]["elpd_loo"], agg[1], rtol=1e-12)
    assert out["config"]["n_high_k"] == agg[4]


def test_rccl_all_reduce_world_size_one():
    """The RCCL path of the sharded reduction (init_process_group("nccl", device_id=...), device-side table, all_reduce,
    device-side merge) in a child process; one rank is all a one-GPU box allows RCCL."""
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from pyloo_amd.engine import get_engine
from pyloo_amd.sharded import all_reduce_aggregates, all_reduce_aggregates_device
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = get_engine(0)
t = torch.empty((5000, 4000), dtype=torch.float64, device=dev)
eng.fill_synthetic(t, seed=77)
agg = eng.psis_loo(t, 190, "psis", 1.0, 0.7, pointwise=False)["agg"]
merged = all_reduce_aggregates(agg, as_tensor=True)
assert merged.is_cuda
timed = all_reduce_aggregates_device(agg, eng)  # the step of the timed multi-GPU loop: pack kernel, all-reduce, merge kernel
torch.cuda.synchronize()
np.testing.assert_allclose(timed.cpu().numpy(), agg.cpu().numpy(), rtol=1e-13)
host = all_reduce_aggregates(agg)
torch.cuda.synchronize()
a, m = agg.cpu().numpy(), merged.cpu().numpy()
np.testing.assert_allclose(m, a, rtol=1e-13)
np.testing.assert_allclose(host, a, rtol=1e-13)
dist.destroy_process_group()
print("rccl ok", a[1])
""" % ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0 and "rccl ok" in proc.stdout, proc.stdout[-1000:] + proc.stderr[-3000:]


def test_chain_generator_and_kernel_record(eng):
    """`pla_fill_synthetic_chains` (bench.py --rows chain_ar1): chain-major AR(1) rows -- lag-1 autocorrelation of the Gaussian
    scores near rho inside a chain, none across the chain boundary, per-chain offsets present -- stay on the fast path, equal
    the oracle, and `pla_engine_last_kernels` names what ran."""
    import torch
    from scipy.special import ndtri

    n, S, chains, rho = 2000, 4000, 4, 0.9
    t = torch.empty((n, S), dtype=torch.float64, device="cuda")
    eng.fill_synthetic_chains(t, seed=11, chains=chains, rho=rho, offset_sd=0.0, k_lo=0.3, k_hi=0.3)
    x = t.cpu().numpy()
    per = S // chains
    # undo the marginal map: ll = -k E + c with E = -log(1 - Phi(z)), k = 0.3, c = -1 - (i mod 7) / 4
    c = -1.0 - (np.arange(n) % 7) * 0.25
    E = -(x - c[:, None]) / 0.3
    z = ndtri(np.clip(1.0 - np.exp(-E), 1e-300, 1 - 1e-16))
    inside = np.mean(z[:, 1:per] * z[:, :per - 1])
    across = np.mean(z[:, per] * z[:, per - 1])
    assert abs(inside - rho) < 0.02 and abs(across) < 0.08, (inside, across)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    eng.fill_synthetic_chains(t, seed=12, chains=chains, rho=rho, offset_sd=0.1)
    res = eng.psis_loo(t, orc.tail_count(S, 1.0), "psis", 1.0, 0.7)
    assert res["agg"][7].item() == 0  # nothing for the general kernel
    idx = np.arange(0, n, 50)
    ref = orc.loo_arrays(t[torch.from_numpy(idx).cuda()].cpu().numpy(), 1.0)
    close(res["diag"].cpu().numpy()[idx], ref["khat"], what="khat")
    close(res["loo_i"].cpu().numpy()[idx], ref["loo_i"], what="loo_i")
    text = eng.last_kernels()
    assert "wave_loo_kernel<double" in text and "fit_rows_stream_kernel" in text, text
    lw, _ = eng.importance_weights(-t[:64], 190, "psis")
    assert "weights" in eng.last_kernels()


def test_aggregate_pack_and_merge_kernels(eng):
    """`pla_aggregate_pack` / `pla_aggregate_merge` (one kernel each around the single all-reduce of a multi-GPU step): three
    ranks' tables packed on one device and summed as the all-reduce would, merged, against the NumPy Chan merge."""
    import torch

    from pyloo_amd.sharded import merge_moment_rows

    rng = np.random.default_rng(4)
    x = rng.normal(-1e4, 0.1, size=9000)
    rows = []
    for lo, hi in ((0, 3000), (3000, 3000), (3000, 9000)):  # (the middle rank holds no observations)
        c = x[lo:hi]
        rows.append([c.size, c.sum(), np.sum((c - c.mean()) ** 2), 2 * c.sum(), c.size % 7, 1, c.min(), 2] if c.size else
                    [0, 0, 0, 0, 0, 0, np.inf, 0])
    world = len(rows)
    total = torch.zeros((world, 8), dtype=torch.float64, device="cuda")
    for r, row in enumerate(rows):
        table = torch.full((world, 8), 7.0, dtype=torch.float64, device="cuda")  # (stale contents must not survive the pack)
        eng.aggregate_pack(torch.tensor(row, dtype=torch.float64, device="cuda"), r, world, table)
        total += table
    out = torch.empty(8, dtype=torch.float64, device="cuda")
    eng.aggregate_merge(total, world, out)
    torch.cuda.synchronize()
    want = merge_moment_rows(np.array(rows, dtype=np.float64))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-13)
    np.testing.assert_allclose(out[2].item() / x.size, np.var(x), rtol=1e-10)


def test_device_path_runs_in_row_blocks(eng):
    """More than 2^20 observations: the device path runs block by block (bounded hand-over buffer, pla_capi.hip).  Rows on
    both sides of the block boundary equal a separate pass over just those rows bit for bit, the aggregate equals the sum of
    the pointwise values, and the row-index flavour (``pla_psis_loo_rows``) walks its index list in the same blocks."""
    import math

    import torch

    n, s = (1 << 20) + 4099, 256
    t = torch.empty((n, s), dtype=torch.float32, device="cuda")
    eng.fill_synthetic(t, seed=0x5EED0011)
    m = orc.tail_count(s, 1.0)
    res = eng.psis_loo(t, m, "psis", 1.0, 0.7)
    lo = (1 << 20) - 7
    part = eng.psis_loo(t[lo:lo + 14], m, "psis", 1.0, 0.7)
    for key in ("diag", "loo_i", "lppd_i"):
        assert torch.equal(res[key][lo:lo + 14], part[key]), key
    loo_i = res["loo_i"].cpu().numpy()
    assert np.isfinite(loo_i).all() and int(res["agg"][0]) == n
    np.testing.assert_allclose(float(res["agg"][1]), math.fsum(loo_i), rtol=1e-12)
    sample = np.arange(0, n, 40009)
    ref = orc.loo_arrays(t[sample].cpu().numpy().astype(np.float64), 1.0)
    np.testing.assert_allclose(loo_i[sample], ref["loo_i"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(res["diag"].cpu().numpy()[sample], ref["khat"], rtol=1e-9, atol=1e-10)
    back = torch.arange(n - 1, -1, -1, device="cuda")
    rev = eng.psis_loo(t, m, "psis", 1.0, 0.7, rows=back)
    assert torch.equal(rev["loo_i"].flip(0), res["loo_i"]) and torch.equal(rev["diag"].flip(0), res["diag"])
