#!/bin/bash
# A/B timing of two builds of the library on the same GPU box (boxes differ by a few percent):
#   ROUNDS=2 tools/ab.sh libA.so libB.so ...   -> kernel ms of the C3 bench for each, alternating
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=${ROUNDS:-2}
for r in $(seq $R); do
  for lib in "$@"; do
    PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/$lib timeout -k 10 200 python $ROOT/bench.py --no-cpu --steps 8 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['roofline']['kernel_ms'],3), 'ms/launch', round(d['ms_per_step'],3), 'ms/step')"
  done
done
