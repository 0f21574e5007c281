#!/bin/bash
# per-kernel durations of any python command of this repo:   bash tools/kstats_cmd.sh tools/bench_weights.py --obs 60000 --draws 20000 --dtype f32
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksc -- python3 $ROOT/"$@" > /tmp/ksc.log 2>&1 || tail -5 /tmp/ksc.log
python3 - <<PY
import csv, glob
for f in glob.glob("/tmp/ksc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-100s calls %4s avg %9.4f ms  %5s %%" % (r["Name"].split("(")[0][:100], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
