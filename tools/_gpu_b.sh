mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/spmm_pmc.sh r03panel > gpurun_out/r03_spmm_pmc_panel.txt 2>&1 || exit 1
bash tools/spmm_pmc.sh r03rows REGT_SPMM_ROWS=1 > gpurun_out/r03_spmm_pmc_rows.txt 2>&1 || exit 1
timeout -k 10 200 python tools/spmm_bench.py > gpurun_out/r03_spmm_bench.txt 2>&1
timeout -k 10 200 python tools/spmm_bench.py wide > gpurun_out/r03_spmm_bench_wide.txt 2>&1
tail -5 gpurun_out/r03_spmm_pmc_panel.txt; tail -5 gpurun_out/r03_spmm_pmc_rows.txt
