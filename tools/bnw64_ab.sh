#!/bin/bash
# A/B of the 64-column fp32 weight-gradient tile (wgrad_kernel<64>: dA0 | dA_r = ds^T [x | L~ x] as one column tile) in the cfg-3 step
for v in "REGT_WGRAD_BNW64=0" "REGT_WGRAD_BNW64=1" "REGT_WGRAD_BNW64=0" "REGT_WGRAD_BNW64=1"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg3 0 20 2>&1 | grep -E "ms/step" | grep -E "mode|wgrad_"
done
