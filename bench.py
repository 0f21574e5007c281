#!/usr/bin/env python3
"""Benchmark of the PSIS-LOO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--obs N_PER_GPU] [--draws S]

A "step" is one fused pass (``pla_psis_loo``: smoothing + loo_i + lppd_i + reductions, plus the
single all-reduce when N > 1) over one device-resident synthetic log-likelihood matrix.  The
default workload is BASELINE.json config C3 -- fp64, S=4000 draws x 1,000,000 observations per
GPU (32 GB), reff=1 -- the configuration the north-star target is quoted on.  Observations are
sharded over ranks (weak scaling: every GPU holds 1e6 rows); ``value`` is whole-job obs/s.

One JSON line is printed by rank 0; it also carries
  roofline      algorithmic bytes (8*S + 24 per observation) / mean kernel time, against 8 TB/s
  cpu_baseline  the NumPy oracle (the reference's per-observation loop restated) on a bounded
                sample of the same matrix, one core; the same sample is the parity check.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def measured_traffic(n_obs, n_draws, dtype):
    """(HBM bytes per LOO pass, file) from the committed rocprofv3 PMC passes of this exact workload (tools/profile.sh ->
    profiles/*traffic*.json: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE, summed over the kernels of the pass),
    or (None, None) when there is none."""
    import glob

    first = os.path.join(ROOT, "profiles", "traffic_latest.json")
    for path in [first] + sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("obs") == n_obs and t.get("draws") == n_draws and t.get("dtype") == dtype:
                return t["hbm_read_bytes"] + t["hbm_write_bytes"], os.path.relpath(path, ROOT)
        except Exception:
            pass
    return None, None


def host_cpu_model():
    """Model string of the host CPU (BASELINE.md section 4 asks for it beside the core count)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform

    return platform.processor() or platform.machine()


def rccl_version(torch):
    """Version of the collective library torch.distributed's "nccl" backend is built on (RCCL on ROCm), or None."""
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:  # (a record field: never worth failing the run for)
        return None


def launch_ranks(n):
    """Start ``n`` ranks of this script through torch.distributed.run (children of a parent that never initialises
    the GPU), pass their output through and return the launcher's exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # fresh children only (this parent has not touched the GPU and never re-executes itself); a launch that hangs -- a rank
    # that never reaches the rendezvous -- ends with a non-zero exit code instead of holding the node
    limit = float(os.environ.get("PYLOO_AMD_BENCH_LAUNCH_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal

        try:
            os.killpg(proc.pid, signal.SIGKILL)  # (the launcher's own process group: exactly what this call started)
        except ProcessLookupError:
            pass
        proc.wait()
        print(f"bench.py: the {n}-rank launch did not finish within {limit:.0f} s", file=sys.stderr)
        return 124


def resolve_workload(args):
    """BASELINE.json presets and the label of the workload: fills args.obs / draws / dtype / seed from --config, labels the
    default workload on a whole node as C4, and returns (label, (heavy_lo, heavy_hi), k_hi) for the generator."""
    heavy = (0.0, 0.0)
    k_hi = 0.60
    if args.config == "C2":
        args.obs, args.draws, args.dtype, args.seed = 100_000, 4000, "f64", 0x5EED0002
    elif args.config in ("C3", "C4"):
        args.obs, args.draws, args.dtype = 1_000_000, 4000, "f64"
        args.seed = 0x5EED0003 if args.config == "C3" else 0x5EED0004
    elif args.config == "C5":  # 70 % rows k ~ U(0.05, 0.5), 30 % rows k ~ U(1.0, 1.3)
        args.obs, args.draws, args.dtype, args.seed = 125_000, 20000, "f32", 0x5EED0005
        heavy, k_hi = (1.0, 1.3), 0.5
    if args.config is None and args.gpus == 8 and (args.draws, args.dtype, args.obs, args.seed) == (4000, "f64", 1_000_000, 0x5EED0003):
        # the default workload on a whole node IS BASELINE.json's C4 (8 M x 4000 f64, observation-sharded): its seed and label
        args.config, args.seed = "C4", 0x5EED0004
    label = args.config or ("C3" if (args.draws, args.dtype, args.obs) == (4000, "f64", 1_000_000) else "custom")
    return label, heavy, k_hi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--obs", type=int, default=1_000_000, help="observations per GPU")
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0003)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--layout", default="draws", choices=["draws", "obs"],
                    help="which index of the matrix is contiguous: the draws of an observation (default, BASELINE's C3) or the "
                         "observations of a draw -- ArviZ's (chain, draw, *obs) as pl.loo(idata) stacks it (loo.py:189), read in place")
    ap.add_argument("--rows", default="iid", choices=["iid", "chain_ar1"],
                    help="iid: BASELINE's generator (exchangeable draws).  chain_ar1: the same marginals as MCMC delivers them -- "
                         "4 chains stacked chain-major, AR(1) with rho = 0.9 in the draw index, per-chain offsets (not a BASELINE "
                         "config: what pl.loo(idata) rows look like to the speculative threshold)")
    ap.add_argument("--config", default=None, choices=["C2", "C3", "C4", "C5"],
                    help="BASELINE.json preset (per-GPU shard): C2 = 1e5 x 4000 f64, C3 / C4 = 1e6 x 4000 f64 per GPU, "
                         "C5 = 125 000 x 20 000 f32 per GPU with 30 %% heavy-tailed rows")
    args = ap.parse_args()
    label, heavy, k_hi = resolve_workload(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has not touched torch or HIP yet, so it only
        # starts N fresh ranks (one per GPU, RCCL over xGMI) and relays rank 0's JSON line and the exit code.
        return launch_ranks(args.gpus)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    # Rehearsal knobs (a one-GPU box cannot run RCCL with two ranks): PYLOO_AMD_BENCH_BACKEND=gloo and
    # PYLOO_AMD_BENCH_DEVICE=0 put every rank on one card and reduce over gloo; the driver's runs use neither.
    backend = os.environ.get("PYLOO_AMD_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("PYLOO_AMD_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus

    from pyloo_amd import _capi
    from pyloo_amd.base import tail_count_for
    from pyloo_amd.engine import get_engine
    from pyloo_amd.sharded import all_reduce_aggregates, all_reduce_aggregates_device

    eng = get_engine(dev_index)
    S, n_local = args.draws, args.obs
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esz = 8 if args.dtype == "f64" else 4
    ll = torch.empty((n_local, S), dtype=tdt, device=dev)
    if args.rows == "chain_ar1":
        eng.fill_synthetic_chains(ll, seed=args.seed, row0=rank * n_local, chains=4, rho=0.9, offset_sd=0.1, k_lo=0.05, k_hi=k_hi)
        label += " rows as chain-major AR(1) chains (rho 0.9, 4 chains, per-chain offsets sd 0.1)"
    else:
        eng.fill_synthetic(ll, seed=args.seed, row0=rank * n_local, k_lo=0.05, k_hi=k_hi, heavy_lo=heavy[0], heavy_hi=heavy[1])
    if args.layout == "obs":
        ll = ll.t().contiguous().t()  # an (n_obs, n_draws) view of an (n_draws, n_obs) buffer: the same numbers, observations fastest
        assert ll.stride(0) == 1
        label += "; observations fastest in memory (the (obs, sample) view of ArviZ's (chain, draw, obs) storage)"
    torch.cuda.synchronize()

    reff = 1.0
    M = tail_count_for(S, reff)
    good_k = min(1 - 1 / np.log10(S), 0.7)

    def step():
        res = eng.psis_loo(ll, M, "psis", 1.0, good_k, pointwise=False, aggregate=True)
        if world > 1:
            # the single collective.  RCCL: pack kernel + all-reduce of the preallocated world x 8 table + merge kernel, all on
            # the device, nothing allocated, no host sync (gloo rehearsals: the same table through host memory)
            if backend == "nccl":
                return all_reduce_aggregates_device(res["agg"], eng)
            return all_reduce_aggregates(res["agg"], as_tensor=True)
        return res["agg"]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # hipEvents bracket every launch of the row kernels on the launch stream, inside the timed region
    # (recorded into a ring and read back after the loop: the host never waits in between)
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        agg = step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_n = eng.kernel_ms()
    f_ms, f_n = eng.first_kernel_ms()
    eng.set_timing(False)
    gave_up = eng.stream_gave_up()  # streamed passes in which the fit kernel stopped waiting for the sweep (0 in a healthy run)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    agg = agg.cpu().numpy() if hasattr(agg, "cpu") else np.asarray(agg)

    # ---- roofline of the LOO pass: wave kernel (selection) + fit kernel + the rows handed to the general kernel,
    #      timed together with HIP events on the launch stream
    kernel_ms = k_ms / max(args.steps, 1)  # per PASS (a pass over more than 2^20 rows is several bracketed launches)
    alg_bytes = n_local * (S * esz + 24.0)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    kernels_text = eng.last_kernels()
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index), "kernel_ms": kernel_ms}
        per_rank = [None] * world
        try:
            dist.all_gather_object(per_rank, mine)
        except Exception:  # (record fields only: the throughput line does not depend on them)
            per_rank = None

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    traffic, traffic_file = measured_traffic(n_local, S, args.dtype)
    if args.layout == "obs":  # (the committed counts are of the draws-fastest pass; this layout's: profiles/r03_col_traffic.txt, per block)
        traffic, traffic_file = None, None
    out = {
        "metric": "psis_loo_observations_per_second",
        "value": world * n_local * args.steps / elapsed,
        "unit": "obs/s",
        "n_gpus": world,
        "ranks": world,
        "backend": (backend if world > 1 else None),
        "rccl_version": rccl_version(torch) if world > 1 and backend == "nccl" else None,
        "devices_seen": ([r["device"] for r in per_rank] if per_rank else [dev_index]),
        "kernel_ms_per_rank": ({"min": min(r["kernel_ms"] for r in per_rank), "max": max(r["kernel_ms"] for r in per_rank)} if per_rank else None),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        # run-time switches in force: what this process's environment sets and what the library itself reads of it
        # (pla_env_overrides; experiment knobs exist only in -DPLA_EXPERIMENT builds, which announce themselves here)
        "env_overrides": sorted(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("PLA_", "PYLOO_AMD_"))),
        "library_env_overrides": _capi.env_overrides(),
        "stream_gave_up": gave_up,
        "config": {
            "workload": f"{label}: synthetic {args.dtype} log_lik S={S} draws x N={n_local} observations per GPU, "
                        f"PSIS-LOO reff=1 (M={M}), device-resident, obs-sharded",
            "obs_per_gpu": n_local,
            "draws": S,
            "seed": hex(args.seed),
            "elpd_loo": float(agg[1]),
            "n_high_k": int(agg[4]),
            "rows_left_to_general_kernel": int(agg[7]),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": (f"{traffic_file} (builder's rocprofv3 --pmc passes of this workload, replayed; not counted in "
                               "this run)") if traffic_file else None,
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "kernels": "whole pass, one event pair on the caller's stream around: " + kernels_text,
        },
    }

    if f_n:
        # the dominant kernel alone (it reads the whole matrix; the fit kernel only sees the <= 250 tail values per observation)
        first_ms = f_ms / f_n
        out["roofline"]["dominant_kernel"] = {
            "name": kernels_text.split("<")[0].split(" ")[0], "kernel_ms": first_ms, "achieved": alg_bytes / (first_ms * 1e-3) / 1e9,
            "frac": alg_bytes / (first_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }

    if not args.no_cpu and world == 1:
        # ---- CPU baseline + parity on a bounded sample of the same matrix (one GPU only: at N > 1 the other ranks would wait
        #      at the closing barrier for it, and the baseline is the same number) -------
        from oracle import psis_oracle as orc

        chunk = max(16, 512 * 4000 // S)
        done, t_cpu = 0, 0.0
        worst = {"khat": 0.0, "loo_i": 0.0, "lppd_i": 0.0}
        cap = min(n_local, 262144)
        # the pass that was timed, once more with its pointwise outputs kept: the same launches on the same matrix, so its
        # aggregates must equal the timed loop's bit for bit -- the rows checked against the oracle below are then rows of
        # the very computation that was timed
        full = eng.psis_loo(ll, M, "psis", 1.0, good_k)
        torch.cuda.synchronize()
        checked_agg = full["agg"].cpu().numpy()
        same_pass = bool(world > 1 or np.array_equal(checked_agg, agg))
        gk, gl, gp = (full[k][:cap].cpu().numpy() for k in ("diag", "loo_i", "lppd_i"))
        while t_cpu < args.cpu_seconds and done + chunk <= cap:
            rows = ll[done:done + chunk].cpu().numpy().astype(np.float64)
            c0 = time.perf_counter()
            ref = orc.loo_pointwise(rows, reff)
            t_cpu += time.perf_counter() - c0
            sl = slice(done, done + chunk)
            for key, got, want in (("khat", gk[sl], ref["diag"]), ("loo_i", gl[sl], ref["loo_i"]), ("lppd_i", gp[sl], ref["lppd_i"])):
                err = np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-2))
                worst[key] = max(worst[key], float(err))
            done += chunk
        out["cpu_baseline"] = {
            "value": done / t_cpu if t_cpu > 0 else None,
            "unit": "obs/s",
            "cores": 1,
            "host_cpu": host_cpu_model(),
            "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"first {done} observations of rank 0's matrix, NumPy oracle (per-observation loop "
                      f"restating pyloo utils.py:171-175 + psis.py:114-160 + loo.py:289-337), {t_cpu:.1f} s",
            "note": "oracle loop, >= reference speed: np.sort where the reference argsorts, no make_ufunc wrapper "
                    "(1.7x the real reference loop on identical rows in the build container, outputs bit-identical)",
        }
        out["parity"] = {"rows": done, "max_rel_err": worst, "tolerance": 1e-6,
                         "checked_pass_equals_timed_pass": same_pass if world == 1 else None}
        # second, clearly labelled CPU line: whole-matrix NumPy calls instead of the loop
        vrows = min(max(64, 2048 * 4000 // S), cap)
        host = ll[:vrows].cpu().numpy().astype(np.float64)
        c0 = time.perf_counter()
        orc.loo_pointwise_vectorised(host, reff)
        out["cpu_baseline_vectorised"] = {
            "value": vrows / (time.perf_counter() - c0), "unit": "obs/s", "cores": 1, "kind": "port",
            "sample": f"first {vrows} observations, batched NumPy restatement (not the reference's loop structure)",
        }
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
