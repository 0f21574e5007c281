#!/bin/bash
# Shader clock, power and temperature while the C3 pass runs back to back (read-only rocm-smi queries beside a long bench run).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
echo "idle:"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8
timeout -k 10 120 python bench.py --no-cpu --steps ${STEPS:-1500} --warmup 3 ${BENCH_ARGS} > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep ${LEAD:-14}
for i in 1 2 3 4 5 6; do
  echo "under load ($i):"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
  sleep 1
done
wait $BP
python -c "
import json; d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1]); print('ms/step', round(d['ms_per_step'],3), 'steps', d['steps'])"
