#!/bin/bash
# start / end of every kernel of the last observations-fastest pass of tools/tile_time.py (rocprofv3 --kernel-trace)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ktrace_tile_tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPS=${REPS:-3} timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/tile_time.py > $OUT/run.log 2>&1 || tail -3 $OUT/run.log
tail -1 $OUT/run.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("VGPR_Count"), r.get("LDS_Block_Size")) for r in rows)
tiles = [k for k in ks if "tile_loo" in k[2]]
t0 = tiles[-1][0] - 30000
for s, e, n, v, l in ks:
    if s >= t0: print("  %9.3f %9.3f ms  %-60s vgpr %s lds %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n, v, l))
PY
