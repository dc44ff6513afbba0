#!/bin/bash
# A/B of non-temporal loads in the fp32 -> bf16 row conversion of the cfg-5 shard step (stage times, same box)
for v in "REGT_CVT_NT=0" "REGT_CVT_NT=1" "REGT_CVT_NT=0" "REGT_CVT_NT=1"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg5shard 2 20 2>&1 | grep -E "ms/step" | grep -E "mode|pack_x|spmm|fused_forward"
done
