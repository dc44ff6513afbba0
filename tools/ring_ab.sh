#!/bin/bash
# A/B of the register-ring depth of the bf16 weight gradient (wgrad_bf16_ring_kernel<D>; 0 = the one-half-slab-ahead kernel)
# inside the cfg-5 shard step: stage times from tools/mode_bench.py, same box, one process per variant.
for v in "REGT_WGRAD_RING=0" "REGT_WGRAD_RING=4" "REGT_WGRAD_RING=6" "REGT_WGRAD_RING=8" "REGT_WGRAD_RING=0" "REGT_WGRAD_RING=6" $RING_AB_EXTRA; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg5shard 2 20 2>&1 | grep -E "ms/step" | grep -E "mode|wgrad_|fused_backward"
done
