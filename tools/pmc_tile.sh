#!/bin/bash
# issue / wait counters of the tile kernel (observations-fastest pass), per observation: bash tools/pmc_tile.sh  (PLA_PIPE=0: the kernel alone)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
N=${OBS:-262144}
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
rm -rf /tmp/pm; OBS=$N REPS=2 rocprofv3 --pmc $set --output-format csv -d /tmp/pm -- python3 $ROOT/tools/tile_time.py > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "tile_loo" in k:
        print(k.split("(")[0][:44], {c: round(sorted(v)[len(v)//2] / $N, 1) for c, v in d.items()}, "per observation", flush=True)
PY
done
