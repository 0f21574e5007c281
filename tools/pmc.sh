#!/bin/bash
# PMC passes for the LOO kernel (separate rocprofv3 runs; no trace domains combined with --pmc)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
ARGS="$ROOT/bench.py --obs ${OBS:-200000} --steps 2 --warmup 1 --no-cpu"
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $ARGS > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        cnt[(k, r["Counter_Name"])] += 1
for k, d in rows.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v/cnt[(k,c)]:.4g}  (per dispatch, {cnt[(k,c)]} dispatches)")
PY
