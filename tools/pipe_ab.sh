#!/bin/bash
# Pipelined split pass, A/B on one GPU box: PLA_PIPE = 0 (kernels back to back), 1 (blocks, slim fit kernel beside the next
# block's wave kernel), 2 (blocks, stand-alone fit kernel) x block sizes; then a kernel trace of the pipelined pass.
#   bash tools/pipe_ab.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
R=${1:-2}
cd "$ROOT"; mkdir -p gpurun_out
one() {  # one bench line -> "ms/step kernel_ms first_kernel_ms elpd"
  env "$@" timeout -k 10 200 python bench.py --no-cpu --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['ms_per_step'],3), 'ms/step', round(r['kernel_ms'],3), 'ms/pass', r.get('dominant_kernel',{}).get('kernel_ms'), d['config']['elpd_loo'], d['config']['n_high_k'])"
}
for r in $(seq $R); do
  for cfg in "PLA_PIPE=0" "PLA_PIPE=1 PLA_PIPE_BLOCK=65536" "PLA_PIPE=1 PLA_PIPE_BLOCK=131072" "PLA_PIPE=1 PLA_PIPE_BLOCK=32768" "PLA_PIPE=2 PLA_PIPE_BLOCK=65536" ${EXTRA_CFG}; do
    echo "$cfg: $(one $cfg)"
  done
done | tee gpurun_out/pipe_ab.txt
