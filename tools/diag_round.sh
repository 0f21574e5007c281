#!/bin/bash
# Diagnostics of the LOO pass on the GPU box: executed instructions + wait breakdown (PMC) for each library given, then the
# phase ablation.  LIBS="libA.so libB.so" bash tools/diag_round.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out
for lib in ${LIBS:-libpyloo_amd.so}; do
  echo "== $lib" | tee -a gpurun_out/diag.txt
  PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/$lib bash tools/kernel_instr.sh 2>&1 | grep -E "wave_loo|fit_rows" | tee -a gpurun_out/diag.txt
  PYLOO_AMD_LIB=$ROOT/pyloo_amd/lib/$lib bash tools/pmc_waits.sh 2>&1 | grep -E "wave_loo|fit_rows" | tee -a gpurun_out/diag.txt
done
if [ -n "$ABLATE" ]; then bash tools/ablate.sh $ABLATE 2>&1 | tee -a gpurun_out/diag.txt; fi
