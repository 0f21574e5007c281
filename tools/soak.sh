#!/bin/bash
# Soak of the streamed passes: the same matrix over and over, every output of every observation compared bit for bit with the
# first pass (what found round 4's spilled-scalar bug: 3-4 observations in 10^4, on some boxes).   bash tools/soak.sh [passes]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-300}
cd "$ROOT"
timeout -k 10 300 python tools/determinism_check.py --quiet --repeat $N || exit $?
timeout -k 10 300 python tools/determinism_check.py --quiet --repeat $N --layout obs || exit $?
timeout -k 10 300 python tools/determinism_check.py --quiet --repeat $((N / 2)) --obs 125000 --draws 20000 --dtype f32 || exit $?
timeout -k 10 300 python tools/determinism_check.py --quiet --repeat $((N / 2)) --obs 500000 --draws 2000 --dtype f32 || exit $?
for i in 1 2; do timeout -k 10 200 python tools/lw_hash.py --obs 500000 --passes 2 || exit $?; done
# (the split weights pass of long rows: rows with tied tail draws may differ between two runs in WHICH tied draw got which quantile,
#  so its k-hat hash -- the last column -- is the one that must repeat)
for i in 1 2; do timeout -k 10 200 python tools/lw_hash.py --obs 40000 --draws 20000 --dtype f32 --passes 2 || exit $?; done
