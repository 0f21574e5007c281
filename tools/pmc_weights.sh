#!/bin/bash
# Where the weights-out pass (pla_importance_weights) spends its time: the same call for PSIS, TIS and SIS (no fit: the
# streaming ceiling of a row held in registers between its read and its write), then issue counters of the PSIS kernel.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OBS=${OBS:-200000}
cd "$ROOT"
for m in psis tis sis; do
  echo "$m: $(timeout -k 10 200 python tools/bench_weights.py --method $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms', round(d['roofline']['frac'],3))")"
done
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
rm -rf /tmp/pw; rocprofv3 --pmc $set --output-format csv -d /tmp/pw -- python3 $ROOT/tools/bench_weights.py --obs $OBS --steps 1 --warmup 1 > /tmp/pw.log 2>&1 || tail -3 /tmp/pw.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pw/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "pla::" in k and "fill" not in k:
        print(k[:60], {c: round(sorted(v)[len(v)//2] / $OBS, 1) for c, v in d.items()})
PY
done
