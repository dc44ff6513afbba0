#!/bin/bash
# PMC passes (HBM bytes, SQ activity) over tools/mode_bench.py for one kernel-name substring (GPU box, repo root):
#   tools/kernel_pmc.sh <tag> <kernel substring> [workload] [mode] [ENV=v ...]
tag=$1; pat=$2; wl=${3:-cfg3}; mode=${4:-0}
shift $(( $# < 4 ? $# : 4 ))
for kv in "$@"; do [[ "$kv" =~ ^[A-Za-z_][A-Za-z0-9_]*= ]] && export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
pass() {
  name=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/kp_${tag}_$name -- python3 tools/mode_bench.py $wl $mode 4 > gpurun_out/kp_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/kp_${tag}_$name.log; return 1; }
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE && \
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES || exit 1
python3 tools/pmc_dump.py "$pat" gpurun_out/kp_${tag}_fetch gpurun_out/kp_${tag}_write gpurun_out/kp_${tag}_sq gpurun_out/kp_${tag}_sq2
python3 tools/pmc_summary.py gpurun_out/kp_${tag}_sq 12
rm -rf gpurun_out/kp_${tag}_fetch gpurun_out/kp_${tag}_write gpurun_out/kp_${tag}_sq gpurun_out/kp_${tag}_sq2
