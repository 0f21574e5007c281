cd $GRAFT_REPO_ROOT
LIB=$GRAFT_REPO_ROOT/pyloo_amd/lib/libpyloo_amd_ablate.so
for r in 1 2; do
for sk in 0 256 128 64 32 4 16; do
  echo "skip=$sk $(PLA_PIPE=0 PLA_SKIP_FIT=1 PYLOO_AMD_LIB=$LIB PLA_DEBUG_SKIP=$sk timeout -k 10 120 python bench.py --obs 1000000 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  pass_ms', round(d['roofline']['kernel_ms'],3), 'first', d['roofline'].get('dominant_kernel',{}).get('kernel_ms'))")"
done; done
echo "production:"; for c in "PLA_PIPE=0" "PLA_PIPE=1"; do echo "$c $(env $c timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'pass_ms', round(d['roofline']['kernel_ms'],3), 'first', d['roofline'].get('dominant_kernel',{}).get('kernel_ms'))")"; done
