#!/bin/bash
# Instructions per observation of every kernel of the LOO pass (PMC, production build): bash tools/kernel_instr.sh
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OBS=${OBS:-100000}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ki
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/ki -- python3 $ROOT/bench.py --obs $OBS --steps 1 --warmup 1 --no-cpu > /tmp/ki.log 2>&1 || tail -3 /tmp/ki.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ki/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "pla::" in k and "fill" not in k:
        print(k[:70], {c: round(sorted(v)[len(v)//2] / $OBS, 1) for c, v in d.items()})
PY
