#!/bin/bash
# rocprofv3 evidence for the bench workload (run on the GPU box through gpurun):
#   1. --kernel-trace --stats            -> per-kernel durations
#   2. --pmc FETCH_SIZE   (own pass)     -> HBM read traffic   (gfx950: counts 64 B per 128-B request: x2)
#   3. --pmc WRITE_SIZE   (own pass)     -> HBM write traffic
# Summaries are written to gpurun_out/prof/summary_<tag>.md|json; copy the ones to keep into profiles/.
set -e
TAG=${1:-r01}
OBS=${OBS:-1000000}
STEPS=${STEPS:-5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
rm -rf $OUT/trace $OUT/fetch $OUT/write   # (a previous run of another workload must not leak into this summary)
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --obs $OBS --steps $STEPS --warmup 2 --no-cpu"
# CONFIG=C5 (etc.): a BASELINE preset instead of --obs (its shapes come from bench.py)
if [ -n "$CONFIG" ]; then ARGS="$ROOT/bench.py --config $CONFIG --steps $STEPS --warmup 2 --no-cpu"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1 || tail -5 $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1 || tail -5 $OUT/write.log
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"; obs = $OBS
stats = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = r
def pmc(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + f"/{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
# per-launch durations from the kernel trace: the bench's 2 warm-up launches run on a cold clock, so the
# average over the $STEPS timed launches is reported next to the all-launch average of --stats
trace = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        trace[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
lines = ["# rocprofv3 summary ($TAG): " + ("bench.py --config $CONFIG" if "$CONFIG" else "bench.py --obs %d" % obs) + " --steps $STEPS --warmup 2 --no-cpu", "",
         "| kernel | calls | avg ms (all launches) | avg ms (last $STEPS = timed) | total ms | % |", "|---|---|---|---|---|---|"]
summary = {"obs": obs, "kernels": {}}
for name, r in sorted(stats.items(), key=lambda kv: -float(kv[1]["TotalDurationNs"])):
    short = name.split("(")[0]
    avg = float(r["AverageNs"]) / 1e6
    d = [x[1] for x in sorted(trace.get(name, []))][-$STEPS:]
    timed = sum(d) / len(d) if d else float("nan")
    lines.append(f"| {short} | {r['Calls']} | {avg:.4f} | {timed:.4f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']} |")
    summary["kernels"][short] = {"calls": int(r["Calls"]), "avg_ms": avg, "avg_ms_timed": timed}
lines += ["", "HBM traffic per launch (KiB counters x 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section):", ""]
for name in fetch:
    if "wave_loo" not in name and "rows_kernel" not in name and "fit_rows" not in name: continue
    short = name.split("(")[0]
    f_kib = sorted(fetch[name])[len(fetch[name]) // 2]
    w_kib = sorted(write.get(name, [0.0]))[len(write.get(name, [0.0])) // 2]
    rd, wr = 2.0 * f_kib * 1024, w_kib * 1024
    lines.append(f"- {short}: FETCH_SIZE {f_kib:.0f} KiB -> read {rd/1e9:.3f} GB (corrected x2), WRITE_SIZE {w_kib:.0f} KiB -> write {wr/1e9:.4f} GB")
    summary["kernels"].setdefault(short, {}).update({"hbm_read_bytes": rd, "hbm_write_bytes": wr, "fetch_size_kib_raw": f_kib})
main = [k for k in summary["kernels"] if ("wave_loo" in k or "rows_kernel" in k or "fit_rows" in k) and "hbm_read_bytes" in summary["kernels"][k]]
if main:
    # the LOO pass is the wave kernel + the fit kernel + the general kernel over the declined rows: bench.py's roofline
    # times the whole pass, so the traffic is summed over its kernels too
    per = {k: {"hbm_read_bytes": summary["kernels"][k]["hbm_read_bytes"], "hbm_write_bytes": summary["kernels"][k]["hbm_write_bytes"],
               "avg_ms_timed": summary["kernels"][k].get("avg_ms_timed")} for k in main}
    # the workload as bench.py itself reports it (its JSON line in trace.log): a --config preset overrides --obs
    shape = {"obs": obs, "draws": int("${DRAWS:-4000}"), "dtype": "${DTYPE:-f64}"}
    for ln in open(out + "/trace.log"):
        if ln.startswith("{") and '"metric"' in ln:
            b = json.loads(ln)
            shape = {"obs": b["config"]["obs_per_gpu"], "draws": b["config"]["draws"], "dtype": b["dtype"]}
    json.dump({**shape, "kernels": per,
               "hbm_read_bytes": sum(v["hbm_read_bytes"] for v in per.values()),
               "hbm_write_bytes": sum(v["hbm_write_bytes"] for v in per.values()),
               "note": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE as is; summed over the kernels of one LOO pass"},
              open(out + "/traffic_$TAG.json", "w"), indent=1)
open(out + "/summary_$TAG.md", "w").write("\n".join(lines) + "\n")
json.dump(summary, open(out + "/summary_$TAG.json", "w"), indent=1)
print("\n".join(lines))
PY
