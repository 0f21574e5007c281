#!/bin/bash
# kernel trace (start / end of every launch) of one bench run: how much of the fit kernels runs beside the wave kernels
#   bash tools/ktrace.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ktrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu --steps 3 --warmup 1 "$@" > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("VGPR_Count"), r.get("LDS_Block_Size")) for r in rows]
ks.sort()
wave = [k for k in ks if "wave_loo" in k[2]]
fit = [k for k in ks if "fit_rows_stream" in k[2] or "fit_rows_slim" in k[2]] or [k for k in ks if "fit_rows" in k[2]]
def total(iv): return sum(e - s for s, e, *_ in iv)
ov = 0
for fs, fe, *_ in fit:
    for ws, we, *_ in wave:
        lo, hi = max(fs, ws), min(fe, we)
        if hi > lo: ov += hi - lo
print("launches: wave", len(wave), "fit", len(fit))
if wave and fit:
    print("wave total ms", total(wave) / 1e6, "avg", total(wave) / 1e6 / len(wave), "vgpr", wave[-1][3], "lds", wave[-1][4])
    print("fit  total ms", total(fit) / 1e6, "avg", total(fit) / 1e6 / len(fit), "vgpr", fit[-1][3], "lds", fit[-1][4])
    print("fit time overlapped with a wave kernel: %.1f %%" % (100.0 * ov / max(total(fit), 1)))
    last = [k for k in ks if k[0] >= wave[-min(len(wave), 2)][0] - 20000]
    t0 = last[0][0]
    for s, e, n, *_ in last[:40]: print("  %9.3f %9.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n))
PY
