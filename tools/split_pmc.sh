#!/bin/bash
# PMC passes over one GEMM (GPU box, repo root): tools/split_pmc.sh <tag> MODE M K N
tag=$1; shift
mkdir -p gpurun_out
run() { name=$1; shift; ctr=$1; shift
  echo "pass $name: $ctr"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/gp_${tag}_$name -- python3 tools/split_pmc.py "$@" > gpurun_out/gp_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/gp_${tag}_$name.log; return 1; }
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "$@" && \
run b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "$@" && \
run c "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "$@"
python tools/pmc_dump.py gemm_flat gpurun_out/gp_${tag}_a gpurun_out/gp_${tag}_b gpurun_out/gp_${tag}_c
