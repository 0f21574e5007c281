#!/bin/bash
# A/B of bench configurations given as env strings, alternating, on one GPU box:
#   ROUNDS=2 bash tools/ab_env.sh "PLA_PIPE=0" "PLA_PIPE=1" "PYLOO_AMD_LIB=/abs/path/lib.so" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
one() { env "$@" timeout -k 10 200 python bench.py --no-cpu --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
dk=r.get('dominant_kernel',{}).get('kernel_ms')
print(round(d['ms_per_step'],3), 'ms/step', round(r['kernel_ms'],3), 'ms/pass', round(dk,3) if dk else dk, 'ms/dominant', d['config']['elpd_loo'], d['config']['n_high_k'], 'slow', d['config']['rows_left_to_general_kernel'], 'gave_up', d.get('stream_gave_up'), d.get('library_env_overrides'))"; }
for r in $(seq ${ROUNDS:-2}); do
  for cfg in "$@"; do echo "$cfg: $(one $cfg)"; done
done
