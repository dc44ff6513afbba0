#!/bin/bash
# A/B of the snapshot packing variants (REGT_PACK) and the aggregation's store policy inside the cfg-3 training step:
# per-stage HIP-event times of pack_x and spmm from tools/mode_bench.py (GPU box, repo root).   tools/pack_ab.sh [steps]
steps=${1:-20}
mkdir -p gpurun_out
for v in "REGT_PACK=0" "REGT_PACK=1" "REGT_PACK=2" "REGT_PACK=3" "REGT_PACK=4" "REGT_PACK=3 REGT_SPMM_NT=0" "REGT_PACK=0" "REGT_PACK=3"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg3 0 $steps 2>&1 | grep -E "ms/step" | grep -E "mode|pack_x|spmm|gemm_regional"
done
