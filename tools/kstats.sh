#!/bin/bash
# per-kernel durations of any python command of this repo:  bash tools/kstats.sh <tag> tools/obs_fastest_cost.py [args]
# (rocprofv3 --kernel-trace --stats; the program itself follows "--": no shell or env wrapper in between)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/kstats_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/"$@" > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | avg ms | total ms | % |"); print("|---|---|---|---|---|")
for r in rows[:14]:
    print(f"| {r['Name'].split('(')[0][:80]} | {r['Calls']} | {float(r['AverageNs'])/1e6:.4f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']} |")
PY
