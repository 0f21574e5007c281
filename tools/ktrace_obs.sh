#!/bin/bash
# per-kernel times of the observations-fastest pass (tools/obs_fastest_cost.py under rocprofv3 --kernel-trace --stats)
#   bash tools/ktrace_obs.sh <tag>      (environment: OBS, PLA_* knobs)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ktrace_obs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/obs_fastest_cost.py > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
tail -1 $OUT/run.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0][:70]
        if any(k in n for k in ("tile_", "fit_rows", "col_", "wave_loo", "slow_rows", "waic")):
            print("%-72s calls %4s  avg %9.3f us  total %9.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
