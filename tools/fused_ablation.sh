#!/bin/bash
# Timing-only ablations of the fused forward kernel (csrc/fused.hip, developer builds with -DREGT_FUSED_ABL=<bits>: 1 no MFMA,
# 2 no gate arithmetic, 4 no transposition through LDS, 8 no per-node sums, 16 weight fragments not loaded, 32 activation stores
# not issued; WRONG results).  Build the variants here, run this script on the GPU box:
#   for v in 1 2 4 8 16 32 48 63; do REGT_LIB_DIR=$PWD/regt-gcn_amd/lib_abl$v REGT_HIPCC_FLAGS=-DREGT_FUSED_ABL=$v python regt-gcn_amd/build.py; done
#   gpurun -- tools/fused_ablation.sh        (regt-gcn_amd/lib_*/ is git-ignored but travels to the box; delete the directories afterwards)
out=gpurun_out/fused_fwd_ablation.txt; : > $out
run() { echo "== $1" >> $out; shift; env "$@" python tools/mode_bench.py cfg5shard 2 10 2>/dev/null | grep -E "mode 2|fused_forward" >> $out; }
run "baseline" A=1
for v in 1 2 4 8 16 32 48 63; do [ -d regt-gcn_amd/lib_abl$v ] && run "REGT_FUSED_ABL=$v" REGT_LIB_DIR=regt-gcn_amd/lib_abl$v; done
run "baseline again" A=1
cat $out
