set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
rm -rf /tmp/pm; rocprofv3 --pmc $set --output-format csv -d /tmp/pm -- python3 $ROOT/bench.py --obs 200000 --steps 1 --warmup 1 --no-cpu > /tmp/pm.log 2>&1 || tail -3 /tmp/pm.log
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "wave_loo" in k or "fit_rows" in k:
        print(k[:40], {c: round(sorted(v)[len(v)//2] / 200000, 1) for c, v in d.items()})
PY
done
