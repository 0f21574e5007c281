#!/bin/bash
# observations-fastest LOO pass against the column kernels' block size (PLA_COL_BLOCK):  bash tools/col_block_ab.sh
for b in 131072 196608 262144; do
  echo "PLA_COL_BLOCK=$b"
  PLA_COL_BLOCK=$b python3 tools/obs_fastest_cost.py 2>/dev/null | cut -c1-330
done
