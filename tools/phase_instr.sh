#!/bin/bash
# VALU instructions per row of the wave kernel by phase: PMC on the ablation build 
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$ROOT/pyloo_amd/lib/alt_ablate.so
(cd "$ROOT" && python -m pyloo_amd.build --alt=ablate -DPLA_WAVE_ABLATE=1 -DPLA_EXPERIMENT)
cd /tmp && export TMPDIR=/tmp
for sk in ${SKIPS:-0 1 4 5 7 8}; do
  rm -rf /tmp/pi_$sk
  PYLOO_AMD_LIB=$LIB PLA_DEBUG_SKIP=$sk rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d /tmp/pi_$sk -- python3 $ROOT/bench.py --obs 100000 --steps 1 --warmup 1 --no-cpu > /tmp/pi_$sk.log 2>&1 || tail -3 /tmp/pi_$sk.log
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pi_$sk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wave_loo" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("skip=$sk", {k: round(sorted(v)[len(v)//2] / 100000) for k, v in acc.items()})
PY
done
