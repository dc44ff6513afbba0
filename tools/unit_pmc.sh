#!/bin/bash
# Which unit is a kernel waiting for?  SQ per-unit activity, LDS and vector-memory path counters over tools/mode_bench.py for one
# kernel-name substring (GPU box, repo root):   tools/unit_pmc.sh <tag> <kernel substring> [workload] [mode]
tag=$1; pat=$2; wl=${3:-cfg5shard}; mode=${4:-2}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
pass() {
  name=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/up_${tag}_$name -- python3 tools/mode_bench.py $wl $mode 4 > gpurun_out/up_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/up_${tag}_$name.log; return 1; }
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU && \
pass c SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES && \
pass d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS && \
pass e TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum && \
pass f TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum || echo "(some pass failed)"
python3 tools/pmc_dump.py "$pat" gpurun_out/up_${tag}_a gpurun_out/up_${tag}_b gpurun_out/up_${tag}_c gpurun_out/up_${tag}_d gpurun_out/up_${tag}_e gpurun_out/up_${tag}_f
rm -rf gpurun_out/up_${tag}_?
