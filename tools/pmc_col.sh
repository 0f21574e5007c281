#!/bin/bash
# HBM traffic of the observations-fastest kernels (separate rocprofv3 --pmc passes):  bash tools/pmc_col.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_col
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  OBS=${OBS:-262144} rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $ROOT/tools/obs_fastest_cost.py > $OUT/$c.log 2>&1 || tail -3 $OUT/$c.log
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    if "col_" in k or "tile_" in k or "fit_rows" in k or "wave_loo" in k:
        med = sorted(v)[len(v) // 2]
        gb = med * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
        print(f"{k:70s} {c:11s} {gb:8.3f} GB per launch (median of {len(v)}; FETCH_SIZE x2 per MI355X_MICROARCH.md)")
PY
