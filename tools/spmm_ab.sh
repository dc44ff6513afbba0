#!/bin/bash
# A/B of aggregation-kernel switches: in the cfg-3 step (tools/mode_bench.py stage times) and alone (tools/spmm_bench.py warm / cold)
steps=${1:-20}
for v in "REGT_SPMM_CSRNT=0" "REGT_SPMM_CSRNT=1" "REGT_SPMM_CSRNT=0" "REGT_SPMM_CSRNT=1"; do
  echo "== in step: $v"
  env $v python3 tools/mode_bench.py cfg3 0 $steps 2>&1 | grep -E "ms/step" | grep -E "mode|pack_x|spmm"
done
for v in "REGT_SPMM_XLD=0" "REGT_SPMM_XLD=416" "REGT_SPMM_XLD=448" "REGT_SPMM_CSRNT=1"; do
  echo "== alone: $v"
  env $v python3 tools/spmm_bench.py 2>&1 | grep -E "^dual  rows=0"
done
echo "== cfg5shard step (bf16 rows; nt loads in the row conversion)"
python3 tools/mode_bench.py cfg5shard 2 20 2>&1 | grep -E "ms/step|cfg5shard" | head -12
