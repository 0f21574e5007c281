#!/bin/bash
# One GPU-box call: the whole -m gpu suite, the draw-order cases against the round-1 sample (A/B), the default bench line.
# Steps after a step that timed out or was killed are skipped.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
step() { "$@"; r=$?; if [ $r -eq 124 ] || [ $r -eq 137 ]; then echo "step killed ($r): $*"; exit $r; fi; return 0; }
rm -f gpurun_out/handover_rates.jsonl gpurun_out/handover_rates_natural.jsonl
step timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/gpu_tests.log 2>&1
tail -5 gpurun_out/gpu_tests.log
if [ -f pyloo_amd/lib/libpyloo_amd_natural.so ]; then
  PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/libpyloo_amd_natural.so PLA_HANDOVER_LOG=handover_rates_natural.jsonl step timeout -k 10 300 python -m pytest tests/test_gpu_robustness.py -m gpu -q > gpurun_out/robust_natural.log 2>&1
  tail -3 gpurun_out/robust_natural.log
fi
step timeout -k 10 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
if [ -n "$AB_LIBS" ]; then
  ROUNDS=${ROUNDS:-2} step bash tools/ab.sh $AB_LIBS > gpurun_out/ab.txt 2>&1
  cat gpurun_out/ab.txt
fi
