#!/bin/bash
# End-of-round evidence in one GPU-box call: full GPU suite, rocprofv3 + PMC passes for C3 and C5, kernel timeline of the
# streamed pass, the bench lines and the side benches.  Everything lands under gpurun_out/final/ (copy what is to be kept
# into profiles/).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
cd "$ROOT"; mkdir -p gpurun_out/final; F=gpurun_out/final
rm -f gpurun_out/handover_rates.jsonl
PLA_HANDOVER_LOG=handover_rates.jsonl timeout -k 10 900 python -m pytest tests -m gpu -q > $F/gpu_tests.log 2>&1; r=$?
tail -3 $F/gpu_tests.log; [ $r -eq 124 ] && exit 124; [ $r -eq 137 ] && exit 137
cp gpurun_out/handover_rates.jsonl $F/${TAG}_handover_rates.jsonl 2>/dev/null
bash tools/profile.sh $TAG > $F/profile_c3.log 2>&1 && cp gpurun_out/prof/summary_$TAG.md $F/${TAG}_rocprof_summary.md && cp gpurun_out/prof/traffic_$TAG.json $F/${TAG}_traffic.json
echo "profile C3 done"
bash tools/ktrace.sh $TAG > $F/${TAG}_kernel_timeline.txt 2>&1; echo "timeline done"
CONFIG=C5 bash tools/profile.sh ${TAG}c5 > $F/profile_c5.log 2>&1 && cp gpurun_out/prof/summary_${TAG}c5.md $F/${TAG}_c5_rocprof_summary.md && cp gpurun_out/prof/traffic_${TAG}c5.json $F/${TAG}_c5_traffic.json
echo "profile C5 done"
timeout -k 10 300 python bench.py > $F/${TAG}_bench.json 2> $F/bench.err; echo "bench C3 rc=$?"
timeout -k 10 300 python bench.py --config C5 > $F/${TAG}_bench_c5.json 2> $F/bench_c5.err; echo "bench C5 rc=$?"
timeout -k 10 300 python bench.py --rows chain_ar1 --no-cpu > $F/${TAG}_bench_chain_ar1.json 2>/dev/null; echo "bench chain_ar1 rc=$?"
PLA_PIPE=0 timeout -k 10 300 python bench.py --no-cpu > $F/${TAG}_bench_back_to_back.json 2>/dev/null; echo "bench PLA_PIPE=0 rc=$?"
timeout -k 10 300 python bench.py --layout obs --no-cpu > $F/${TAG}_bench_layout_obs.json 2>/dev/null; echo "bench --layout obs rc=$?"
timeout -k 10 200 python tools/bench_weights.py 2>/dev/null > $F/${TAG}_bench_weights.json
timeout -k 10 200 python tools/bench_weights.py --obs 60000 --draws 20000 --dtype f32 2>/dev/null > $F/${TAG}_bench_weights_s20000_f32.json
timeout -k 10 200 python tools/bench_is.py --method tis 2>/dev/null > $F/${TAG}_bench_tis.json
timeout -k 10 200 python tools/bench_is.py --method sis 2>/dev/null > $F/${TAG}_bench_sis.json
timeout -k 10 200 python tools/bench_waic.py 2>/dev/null > $F/${TAG}_bench_waic.json
timeout -k 10 200 python tools/bench_e_loo.py 2>/dev/null > $F/${TAG}_bench_e_loo.json
timeout -k 10 200 python tools/obs_fastest_cost.py 2>/dev/null > $F/${TAG}_bench_obs_fastest.json
timeout -k 10 200 python tools/tile_time.py 2>/dev/null | tail -1 > $F/${TAG}_bench_obs_fastest_tile_streamed.json
PLA_PIPE=0 timeout -k 10 200 python tools/tile_time.py 2>/dev/null | tail -1 > $F/${TAG}_bench_obs_fastest_tile_back_to_back.json
# (PLA_NO_TILE is an experiment knob: compiled into an alternative build only)
python -m pyloo_amd.build --alt=experiment -DPLA_EXPERIMENT > /dev/null 2>&1 && \
  PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/alt_experiment.so PLA_NO_TILE=1 timeout -k 10 200 python tools/tile_time.py 2>/dev/null | tail -1 > $F/${TAG}_bench_obs_fastest_lane_per_observation.json
timeout -k 10 200 python tools/waic_col_time.py 2>/dev/null | tail -1 > $F/${TAG}_bench_waic_obs_fastest.json
bash tools/ktrace_tile_timeline.sh 2>&1 | grep -v "^W2026" > $F/${TAG}_tile_kernel_timeline.txt; echo "tile timeline done"
OBS=262144 timeout -k 10 600 bash tools/pmc_col.sh > $F/${TAG}_col_traffic.txt 2>&1; echo "tile traffic done"
for f in $F/${TAG}_bench*.json; do echo "== $f"; cut -c1-420 $f; done
