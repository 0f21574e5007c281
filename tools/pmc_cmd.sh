#!/bin/bash
# Issue counters per unit of work of every pla:: kernel of any python command of this repo:
#   UNITS=200000 bash tools/pmc_cmd.sh tools/bench_e_loo.py
#   UNITS=125000 bash tools/pmc_cmd.sh bench.py --config C5 --steps 1 --warmup 1 --no-cpu
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
UNITS=${UNITS:-200000}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
rm -rf /tmp/pc; rocprofv3 --pmc $set --output-format csv -d /tmp/pc -- python3 $ROOT/"$@" > /tmp/pc.log 2>&1 || tail -3 /tmp/pc.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "pla::" in k and "fill" not in k and "zero" not in k:
        print(k[:64], {c: round(sorted(v)[len(v)//2] / $UNITS, 1) for c, v in d.items()})
PY
done
