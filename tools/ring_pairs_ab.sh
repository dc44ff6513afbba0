#!/bin/bash
# bf16 weight gradients of the cfg-5 shard step: ring depth x tile rows x paired launches (dhp^T [q | A_hat x], dzr^T [h | A_hat x]);
# stage times from tools/mode_bench.py, one process per variant, same box.   tools/ring_pairs_ab.sh ["ENV=v ENV=v" ...]
variants=("REGT_WGRAD_RING=0 REGT_WGRAD_PAIRS=0" "REGT_WGRAD_RING=6 REGT_WGRAD_PAIRS=0" "REGT_WGRAD_RING=6 REGT_WGRAD_PAIRS=1" "REGT_WGRAD_RING=8 REGT_WGRAD_PAIRS=1" "REGT_WGRAD_RING=4 REGT_WGRAD_PAIRS=1" "REGT_WGRAD_RING=6 REGT_WGRAD_PAIRS=1 REGT_WGRAD_TILE=256" "REGT_WGRAD_RING=6 REGT_WGRAD_PAIRS=0 REGT_WGRAD_TILE=256" "REGT_WGRAD_RING=6 REGT_WGRAD_PAIRS=1")
[ $# -gt 0 ] && variants=("$@")
for v in "${variants[@]}"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg5shard 2 20 2>&1 | grep -E "ms/step" | grep -E "mode|wgrad_|fused_backward"
done
