#!/bin/bash
# PMC passes over tools/spmm_bench.py (run on the GPU box from the repo root): tools/spmm_pmc.sh <tag> [env assignments...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
run() {  # name, counters...
  name=$1; shift
  echo "pass $name: $*"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/sp_${tag}_$name -- python3 tools/spmm_bench.py > gpurun_out/sp_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/sp_${tag}_$name.log; return 1; }
}
run a WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && run b FETCH_SIZE && run c TA_BUSY_avr TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR && run d SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python tools/pmc_dump.py spmm_dual gpurun_out/sp_${tag}_a gpurun_out/sp_${tag}_b gpurun_out/sp_${tag}_c gpurun_out/sp_${tag}_d
