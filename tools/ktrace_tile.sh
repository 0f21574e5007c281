#!/bin/bash
# time of tile_loo_kernel alone (rocprofv3 --kernel-trace --stats around tools/tile_time.py) for a list of library builds
#   bash tools/ktrace_tile.sh default a1 a2 ...      (names of pyloo_amd/lib/alt_<name>.so; "default" = the in-tree build)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for NAME in "$@"; do
  OUT=$ROOT/gpurun_out/ktrace_tile_$NAME
  rm -rf $OUT; mkdir -p $OUT
  if [ "$NAME" = default ]; then unset PYLOO_AMD_LIB; else export PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/alt_$NAME.so; fi
  REPS=${REPS:-3} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/tile_time.py > $OUT/run.log 2>&1 || tail -3 $OUT/run.log
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].split("(")[0]
        if any(k in n for k in ("tile_", "fit_rows", "slow_rows", "reduce", "wave_loo")):
            print("%-10s %-40s calls %3s  avg %9.3f us" % ("$NAME", n[:40], r["Calls"], float(r["AverageNs"]) / 1e3), flush=True)
PY
done
