#!/bin/bash
# A/B of the generated-operand candidate data gradient inside the cfg-3 step (fp32): stage times from tools/mode_bench.py
for v in "REGT_DGRAD1_GEN=0" "REGT_DGRAD1_GEN=1" "REGT_DGRAD1_GEN=0" "REGT_DGRAD1_GEN=1"; do
  echo "== $v"
  env $v python3 tools/mode_bench.py cfg3 0 20 2>&1 | grep -E "ms/step" | grep -E "mode|cell_bwd|dgrad_candidate|dgrad_gates|wgrad_Uh|wgrad_Gh"
done
