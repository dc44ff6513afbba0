#!/bin/bash
# A/B of the chunk count of the wide fp32 weight gradients (dUh, dUzr) in the cfg-3 step: ~128 chunks (0) against 1 or 2 full waves
for v in "REGT_WGRAD_WAVE32=0" "REGT_WGRAD_WAVE32=1" "REGT_WGRAD_WAVE32=2" "REGT_WGRAD_WAVE32=0" "REGT_WGRAD_WAVE32=1" "REGT_WGRAD_WAVE32=2"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg3 0 20 2>&1 | grep -E "ms/step" | grep -E "mode|wgrad_U|wgrad_reduce"
done
