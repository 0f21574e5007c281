#!/usr/bin/env python3
"""Order-sensitive hashes of the whole output of the weights pass (pla_importance_weights), for comparing two builds of the
library bit for bit on matrices too large to carry around:

    python tools/lw_hash.py [--obs N] [--draws S] [--dtype f64|f32] [--passes K]
    PYLOO_AMD_LIB=/path/to/other.so python tools/lw_hash.py ...

Prints one line per pass: sum of the bit patterns and sum of bit pattern x position, both modulo 2^64."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obs", type=int, default=200_000)
    ap.add_argument("--draws", type=int, default=4000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--passes", type=int, default=3)
    args = ap.parse_args()
    import torch

    from pyloo_amd.base import tail_count_for
    from pyloo_amd.engine import get_engine

    eng = get_engine(0)
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ll = torch.empty((args.obs, args.draws), dtype=tdt, device="cuda")
    M = tail_count_for(args.draws, 1.0)
    pos = None
    for p in range(args.passes):
        eng.fill_synthetic(ll, seed=0x5EED0100 + p, k_lo=0.05, k_hi=0.9)
        ll.neg_()
        lw, k = eng.importance_weights(ll, M, "psis")
        torch.cuda.synchronize()
        bits = lw.view(torch.int64 if args.dtype == "f64" else torch.int32).to(torch.int64).reshape(-1)
        if pos is None:
            pos = torch.arange(bits.numel(), dtype=torch.int64, device="cuda") | 1
        print(p, int(bits.sum().item()) & (2**64 - 1), int((bits * pos).sum().item()) & (2**64 - 1), int(k.view(torch.int64).sum().item()) & (2**64 - 1))
        del bits, lw


if __name__ == "__main__":
    main()
