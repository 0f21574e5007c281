"""Observation-sharded multi-GPU reduction.

Every rank runs the fused pass on its own contiguous block of observations; the only
communication is ONE all-reduce (RCCL over xGMI when the backend is "nccl", gloo on CPU in
tests) of a zero-padded ``world x 8`` table in which each rank fills its own row with
``[n, sum loo_i, M2 about its own mean, sum lppd_i, #k>good_k, #non-finite k, min diag, slow]``.
Summing the table is an all-gather of the per-rank moments; they are then merged with the
pairwise update of Chan, Golub & LeVeque (1983), which has no cancellation, so ``se`` and
``p_loo_se`` equal the single-device ``np.var`` result to rounding (loo.py:327,340).
"""

import numpy as np

from ._capi import AGG_COUNT, AGG_M2_LOO, AGG_MIN_DIAG, AGG_N, AGG_N_HIGH, AGG_N_NONFINITE, AGG_N_SLOW, AGG_SUM_LOO, AGG_SUM_LPPD


def shard_bounds(n_obs, world_size, rank):
    """Contiguous block ``[lo, hi)`` of observations owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(int(n_obs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_moment_rows(table):
    """Combine per-rank aggregate rows (``world x AGG_COUNT``) into one aggregate vector."""
    table = np.asarray(table, dtype=np.float64).reshape(-1, AGG_COUNT)
    out = np.zeros(AGG_COUNT)
    out[AGG_MIN_DIAG] = np.inf
    n, mean, m2 = 0.0, 0.0, 0.0
    for row in table:
        nb = row[AGG_N]
        if nb == 0:
            continue
        mb = row[AGG_SUM_LOO] / nb
        if n == 0:
            n, mean, m2 = nb, mb, row[AGG_M2_LOO]
        else:
            delta = mb - mean
            tot = n + nb
            m2 = m2 + row[AGG_M2_LOO] + delta * delta * n * nb / tot
            mean = mean + delta * nb / tot
            n = tot
        out[AGG_SUM_LOO] += row[AGG_SUM_LOO]
        out[AGG_SUM_LPPD] += row[AGG_SUM_LPPD]
        out[AGG_N_HIGH] += row[AGG_N_HIGH]
        out[AGG_N_NONFINITE] += row[AGG_N_NONFINITE]
        out[AGG_N_SLOW] += row[AGG_N_SLOW]
        out[AGG_MIN_DIAG] = min(out[AGG_MIN_DIAG], row[AGG_MIN_DIAG])
    out[AGG_N] = n
    out[AGG_M2_LOO] = m2
    return out


def merge_moment_rows_tensor(table):
    """:func:`merge_moment_rows` on a torch tensor (any device), without a host round trip: the pooled form
    ``M2 = sum M2_r + sum n_r (mean_r - mean)^2`` of the same update (a sum of non-negative terms)."""
    import torch

    n_r = table[:, AGG_N]
    n = n_r.sum()
    safe = torch.where(n_r > 0, n_r, torch.ones_like(n_r))
    mean_r = table[:, AGG_SUM_LOO] / safe
    mean = table[:, AGG_SUM_LOO].sum() / torch.clamp(n, min=1.0)
    out = torch.zeros(AGG_COUNT, dtype=table.dtype, device=table.device)
    out[AGG_N] = n
    out[AGG_SUM_LOO] = table[:, AGG_SUM_LOO].sum()
    out[AGG_M2_LOO] = table[:, AGG_M2_LOO].sum() + (n_r * (mean_r - mean) ** 2).sum()
    out[AGG_SUM_LPPD] = table[:, AGG_SUM_LPPD].sum()
    out[AGG_N_HIGH] = table[:, AGG_N_HIGH].sum()
    out[AGG_N_NONFINITE] = table[:, AGG_N_NONFINITE].sum()
    out[AGG_N_SLOW] = table[:, AGG_N_SLOW].sum()
    inf = torch.full_like(n_r, float("inf"))
    out[AGG_MIN_DIAG] = torch.where(n_r > 0, table[:, AGG_MIN_DIAG], inf).min()
    return out


_TABLES = {}  # (device, world, group) -> (table, out): allocated once, so that a timed step allocates nothing


def all_reduce_aggregates_device(agg, engine, group=None):
    """The multi-GPU step of a timed loop: ``agg`` is this rank's aggregate vector on the GPU; the merged vector comes back
    as a CUDA tensor, identical on every rank.  One step is exactly: one pack kernel (``pla_aggregate_pack``: this rank's row
    of a preallocated ``world x 8`` table, the other rows zero), ONE all-reduce of that table (RCCL over xGMI) and one merge
    kernel (``pla_aggregate_merge``); nothing synchronises with the host and nothing is allocated.

    The returned tensor is the preallocated result buffer of (device, world, group): it is valid until the NEXT call with the
    same three, which overwrites it -- ``.clone()`` it to keep a step's result beyond that."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    key = (agg.device, world, id(group) if group is not None else None)
    if key not in _TABLES:
        _TABLES[key] = (torch.zeros((world, AGG_COUNT), dtype=torch.float64, device=agg.device),
                        torch.zeros(AGG_COUNT, dtype=torch.float64, device=agg.device))
    table, out = _TABLES[key]
    engine.aggregate_pack(agg, rank, world, table)
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)  # the single collective
    engine.aggregate_merge(table, world, out)
    return out


def all_reduce_aggregates(agg, group=None, as_tensor=False):
    """``agg``: this rank's aggregate vector (CUDA tensor, CPU tensor or ndarray).
    Returns the merged aggregate vector, identical on every rank: a NumPy array, or with ``as_tensor`` a
    tensor on the device the collective ran on (RCCL: the GPU; nothing synchronises with the host)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        if as_tensor and hasattr(agg, "detach"):
            return merge_moment_rows_tensor(agg.detach().reshape(1, AGG_COUNT))
        a = agg.detach().cpu().numpy() if hasattr(agg, "detach") else np.asarray(agg)
        return merge_moment_rows(a[None, :])
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    t = agg if hasattr(agg, "detach") else torch.as_tensor(np.asarray(agg), dtype=torch.float64)
    # RCCL reduces device tensors in place; gloo (CPU tests, rehearsals) wants host tensors
    where = t.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    table = torch.zeros((world, AGG_COUNT), dtype=torch.float64, device=where)
    table[rank] = t.to(device=where, dtype=torch.float64)
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)  # the single collective
    if as_tensor:
        return merge_moment_rows_tensor(table)
    return merge_moment_rows(table.cpu().numpy())
