#!/bin/bash
# an alternative build of the library for A/B runs:  bash tools/build_alt.sh <name> -DPLA_X=..   ->  pyloo_amd/lib/alt_<name>.so
# (use with PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/alt_<name>.so; *.so is git-ignored and travels with gpurun)
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -mllvm -disable-machine-licm "$@" -I$ROOT/include \
  -o $ROOT/pyloo_amd/lib/alt_$NAME.so $ROOT/pyloo_amd/csrc/pla_kernels.hip $ROOT/pyloo_amd/csrc/pla_capi.hip && echo built alt_$NAME.so
