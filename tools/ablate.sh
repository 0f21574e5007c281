#!/bin/bash
# phase ablation of the fast kernel (profiling aid): PLA_DEBUG_SKIP bits
#  1: no exp evaluation   2: no histogram atomics (invalid results; with 4 only)   4: stop after the sweep
#  8: no GPD fit / smoothing
for sk in "$@"; do
  echo "skip=$sk"; PLA_DEBUG_SKIP=$sk timeout -k 10 120 python bench.py --obs 200000 --steps 5 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  kernel_ms', round(d['roofline']['kernel_ms'],3), 'GB/s', round(d['roofline']['achieved'],1))"
done
