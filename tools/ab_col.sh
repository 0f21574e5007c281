#!/bin/bash
# A/B of builds on the observations-fastest LOO pass (same box):  tools/ab_col.sh libA.so libB.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for r in $(seq ${ROUNDS:-2}); do
  for lib in "$@"; do
    PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/$lib timeout -k 10 200 python3 $ROOT/tools/obs_fastest_cost.py 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'obs-fastest', round(d['loo_obs_fastest_ms'],3), 'ms; draws-fastest', round(d['loo_draws_fastest_ms'],3), 'rows to general', d['rows_to_general_kernel'])"
  done
done
