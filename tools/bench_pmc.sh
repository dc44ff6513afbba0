#!/bin/bash
# rocprofv3 passes over bench.py (GPU box, repo root): kernel-trace stats + three PMC passes (counters in their own runs).
#   tools/bench_pmc.sh <tag> [bench.py args...]      e.g.  tools/bench_pmc.sh cfg3 --workload cfg3
# Writes gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_hbm_summary.txt, <tag>_pmc_sq_summary.txt, <tag>_bench.json;
# copy the ones to be judged into profiles/ (bench.py reads profiles/r05_<workload>_pmc_hbm_summary.txt for `traffic`).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
mkdir -p gpurun_out
common="--no-cpu-baseline --no-split-leg --no-tpims-leg --no-cfg5-leg"
python3 bench.py "$@" --steps 20 --warmup 3 $common > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "plain bench failed"; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -- python3 bench.py "$@" --steps 12 --warmup 3 $common > gpurun_out/${tag}_kt.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 gpurun_out/${tag}_kt.log; exit 1; }
cp gpurun_out/${tag}_kt/*/*kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv
echo "kernel trace done"
pass() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/${tag}_$name -- python3 bench.py "${BARGS[@]}" --steps 4 --warmup 2 $common > gpurun_out/${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/${tag}_$name.log; return 1; }
  echo "pass $name done"
}
BARGS=("$@")
pass fetch FETCH_SIZE && pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum && pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE || exit 1
python3 tools/pmc_summary.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write 16 > gpurun_out/${tag}_pmc_hbm_summary.txt
python3 tools/pmc_summary.py gpurun_out/${tag}_sq 16 > gpurun_out/${tag}_pmc_sq_summary.txt
rm -rf gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_sq
head -3 gpurun_out/${tag}_pmc_hbm_summary.txt
