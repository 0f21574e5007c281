#!/bin/bash
# an alternative build of the library for A/B runs:  bash tools/build_alt.sh <name> -DPLA_X=..   ->  pyloo_amd/lib/alt_<name>.so
# (use with PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/alt_<name>.so; *.so is git-ignored and travels with gpurun)
# -DPLA_EXPERIMENT switches the experiment knobs on (csrc/pla_launch.h): PLA_DEBUG_SKIP, PLA_FUSED, PLA_WAVE_PRIO, ...
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" && python -m pyloo_amd.build --alt=$NAME "$@" && echo built alt_$NAME.so
