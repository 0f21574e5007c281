#!/bin/bash
# tests + A/B + C5 + split ablation in one GPU-box call
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/handover_rates.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; r=$?
tail -15 gpurun_out/gpu_tests.log; [ $r -eq 124 ] && exit 124; [ $r -eq 137 ] && exit 137
cat gpurun_out/handover_rates.jsonl
ROUNDS=${ROUNDS:-2} bash tools/ab.sh ${AB_LIBS:-libpyloo_amd.so libpyloo_amd_fitsorts0.so} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab.txt
for f in 0 1; do
  echo "C5 shard, PLA_FUSED=$f"
  PLA_FUSED=$f timeout -k 10 200 python bench.py --config C5 --no-cpu --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['achieved']), 'GB/s', d['roofline'].get('dominant_kernel'), d['config']['n_high_k'])"
done | tee gpurun_out/c5.txt
[ -n "$ABLATE" ] && OBS=1000000 bash tools/ablate.sh $ABLATE 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ablate_split.txt
