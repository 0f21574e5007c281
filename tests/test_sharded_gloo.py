"""Observation sharding across ranks on CPU: two processes, ``gloo``.

Every rank runs the per-observation pass on ITS OWN block of observations (here through the
oracle-backed stand-in engine: no GPU in this suite) and the aggregates are merged with the one
all-reduce of ``pyloo_amd.sharded``.  The merged numbers must equal the single-process result of
the reference arithmetic on the whole matrix."""

import os
import socket
import sys

import numpy as np
import pytest

from oracle import psis_oracle as orc
from pyloo_amd.sharded import merge_moment_rows, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            blocks = [shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(b[1] == blocks[i + 1][0] for i, b in enumerate(blocks[:-1]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_moment_merge_equals_numpy_var():
    rng = np.random.default_rng(9)
    x = rng.normal(-1e4, 0.1, size=10_000)  # |mean| >> sd: naive sum-of-squares would lose 1e-6
    rows = []
    for lo, hi in ((0, 10), (10, 10), (10, 4000), (4000, 10_000)):
        c = x[lo:hi]
        rows.append([c.size, c.sum(), np.sum((c - c.mean()) ** 2) if c.size else 0.0, c.sum() * 2, 1 if c.size else 0, 0, 0.5, 0])
    out = merge_moment_rows(np.array(rows))
    assert out[0] == x.size
    np.testing.assert_allclose(out[1], x.sum(), rtol=1e-14)
    np.testing.assert_allclose(out[2] / x.size, np.var(x), rtol=1e-10)
    assert out[4] == 3 and out[6] == 0.5


def test_tensor_merge_equals_numpy_merge():
    """The device-side merge used inside the multi-GPU timed loop (no host sync) against the Chan update."""
    import torch

    from pyloo_amd.sharded import merge_moment_rows_tensor

    rng = np.random.default_rng(10)
    x = rng.normal(-1e4, 0.1, size=9_000)
    rows = []
    for lo, hi in ((0, 0), (0, 3000), (3000, 3001), (3001, 9000)):
        c = x[lo:hi]
        rows.append([c.size, c.sum(), np.sum((c - c.mean()) ** 2) if c.size else 0.0, c.sum() * 2, c.size % 7, 1, c.min(), 2] if c.size else [0.0] * 8)
    table = np.array(rows)
    want = merge_moment_rows(table)
    got = merge_moment_rows_tensor(torch.from_numpy(table)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-13)
    np.testing.assert_allclose(got[2] / x.size, np.var(x), rtol=1e-10)


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks through torch.distributed.run on the loopback
    interface and relays the exit code; the parent never imports torch."""
    import importlib.util
    import subprocess

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeProc:
        pid = 0

        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"], seen["kw"] = cmd, env, kw

        def wait(self, timeout=None):
            seen["timeout"] = timeout
            if seen.get("hang") and timeout is not None:
                seen["hang"] = False
                raise subprocess.TimeoutExpired(seen["cmd"], timeout)
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(os, "killpg", lambda pid, sig: seen.setdefault("killed", (pid, sig)))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    before = "torch" in sys.modules
    assert bench.main() == 7
    assert ("torch" in sys.modules) == before
    assert seen["timeout"] and seen["kw"].get("start_new_session")  # bounded wait, a process group of its own
    # a launch that never finishes: its process group is killed and the exit code is non-zero
    seen["hang"] = True
    assert bench.main() == 124 and "killed" in seen
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ


def _worker(rank, world, port, ll, reff, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import warnings

    import torch.distributed as dist

    from fake_engine import OracleEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = OracleEngine()
        importlib.import_module("pyloo_amd.loo").get_engine = lambda device=None: eng
        import pyloo_amd as pl

        lo, hi = shard_bounds(ll.shape[0], world, rank)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = pl.loo_from_matrix(ll[lo:hi], reff=reff, pointwise=True, distributed=True)
        out_q.put((rank, {k: float(res[k]) for k in ("elpd_loo", "se", "p_loo", "p_loo_se", "looic", "looic_se")},
                   int(res["n_data_points"]), bool(res["warning"]), np.asarray(res["loo_i"]).shape[0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_obs,world", [(37, 2), (64, 2), (41, 8)])
def test_ranks_match_single_process(n_obs, world):
    """World size 2, and the node's 8 (uneven blocks: 41 observations over 8 ranks): the merged aggregates on every rank."""
    import torch.multiprocessing as mp

    rng = np.random.default_rng(n_obs)
    k = rng.uniform(0.1, 1.0, size=n_obs)
    ll = -k[:, None] * rng.exponential(size=(n_obs, 600)) - 2.0
    reff = 0.8
    want = orc.loo_arrays(ll, reff)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ll, reff, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = 0
    for rank, vals, n_total, warn, n_local in got:
        assert n_total == n_obs and warn == (want["n_high_k"] > 0)
        for key, v in vals.items():
            np.testing.assert_allclose(v, want[key], rtol=1e-11, err_msg=f"rank {rank} {key}")
        sizes += n_local
    assert sizes == n_obs  # pointwise outputs stay sharded


def test_a_launch_that_hangs_ends_with_exit_code_124():
    """The real launcher, no stand-in: `python bench.py --gpus 2` whose ranks cannot finish inside the launch timeout (they
    are still importing torch when it expires) -- the parent kills the launcher's process group (fresh children only: it
    never touched the GPU itself) and exits with 124 instead of holding the node."""
    import subprocess
    import time

    env = dict(os.environ, PYLOO_AMD_BENCH_LAUNCH_TIMEOUT="0.2", PYLOO_AMD_BENCH_BACKEND="gloo", PYLOO_AMD_BENCH_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--obs", "1000", "--steps", "1", "--warmup", "0",
                          "--no-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 124, (out.returncode, out.stderr[-400:])
    assert "did not finish within" in out.stderr and time.time() - t0 < 60
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no record line from a launch that was cut off


def test_default_node_run_is_labelled_c4():
    """`bench.py --gpus 8` with the default workload is BASELINE.json's C4: its label and its seed --
    through the function main() resolves its workload with, for the commands the driver runs at 1, 2, 4 and 8 GPUs."""
    import argparse
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def resolve(gpus, **over):
        a = argparse.Namespace(gpus=gpus, obs=1_000_000, draws=4000, dtype="f64", seed=0x5EED0003, config=None)
        for k, v in over.items():
            setattr(a, k, v)
        label, heavy, k_hi = bench.resolve_workload(a)
        return a, label, heavy, k_hi

    a, label, heavy, k_hi = resolve(8)
    assert (label, a.config, a.seed, a.obs, a.draws, a.dtype) == ("C4", "C4", 0x5EED0004, 1_000_000, 4000, "f64")
    for g in (1, 2, 4):
        a, label, _, _ = resolve(g)
        assert (label, a.config, a.seed) == ("C3", None, 0x5EED0003), g
    a, label, heavy, k_hi = resolve(8, config="C5")
    assert (label, a.obs, a.draws, a.dtype, a.seed, heavy, k_hi) == ("C5", 125_000, 20000, "f32", 0x5EED0005, (1.0, 1.3), 0.5)
    a, label, _, _ = resolve(8, obs=4000)   # a reduced rehearsal is not C4
    assert (label, a.config, a.seed) == ("custom", None, 0x5EED0003)
    assert isinstance(bench.host_cpu_model(), str) and bench.host_cpu_model()


def test_device_side_merge_kernel_contract():
    """`pla_aggregate_pack` / `pla_aggregate_merge` are declared, exported and bound (one kernel each around the single
    all-reduce of a timed multi-GPU step); their arithmetic is the Chan merge tested above, checked on the GPU in
    tests/test_gpu_baseline_configs.py."""
    from pyloo_amd import _capi

    assert {"pla_aggregate_pack", "pla_aggregate_merge"} <= set(_capi.SYMBOLS)
    lib = _capi.load_library()
    assert lib.pla_aggregate_pack(None, None, 0, 1, None, None) < 0  # argument errors are status codes
    assert lib.pla_aggregate_merge(None, None, 0, None, None) < 0
