#!/bin/bash
# Phase ablation of the wave kernel (profiling aid).  Uses a separate build with PLA_WAVE_ABLATE=1;
# PLA_DEBUG_SKIP bits: 1 no exponentials, 2 no histogram atomics, 4 stop after the sweep,
# 8 no GPD fit / smoothing, 16 statistics + threshold only.  Results of ablated runs are meaningless; only the times matter.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/pyloo_amd/lib/alt_ablate.so
if [ ! -f "$LIB" ] || [ "$ROOT/pyloo_amd/csrc/pla_wave.h" -nt "$LIB" ]; then
  (cd "$ROOT" && python -m pyloo_amd.build --alt=ablate -DPLA_WAVE_ABLATE=1 -DPLA_EXPERIMENT)
fi
for sk in "$@"; do
  echo "skip=$sk"; PYLOO_AMD_LIB=$LIB PLA_DEBUG_SKIP=$sk timeout -k 10 120 python "$ROOT/bench.py" --obs ${OBS:-200000} --steps 5 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  kernel_ms', round(d['roofline']['kernel_ms'],3), 'GB/s', round(d['roofline']['achieved'],1))"
done
