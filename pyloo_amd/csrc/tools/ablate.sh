for sk in 0 1 2 3 4 7 8 16 24; do
  echo "skip=$sk"; PLA_DEBUG_SKIP=$sk timeout -k 10 120 python bench.py --obs 200000 --steps 5 --warmup 2 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  kernel_ms', round(d['roofline']['kernel_ms'],3), 'GB/s', round(d['roofline']['achieved'],1))"
done
