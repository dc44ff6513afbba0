mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3l_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3l_tests.log
timeout -k 10 200 python bench.py --workload cfg5shard --no-cpu-baseline --no-split-leg --steps 20 > gpurun_out/r3l_cfg5.json 2> gpurun_out/r3l_cfg5.err; python tools/show_bench.py gpurun_out/r3l_cfg5.json > gpurun_out/r3l_cfg5.txt; head -26 gpurun_out/r3l_cfg5.txt | grep -v roofline
timeout -k 10 300 python bench.py --no-cpu-baseline --no-split-leg --no-tpims-leg --steps 20 > gpurun_out/r3l_cfg3.json 2> gpurun_out/r3l_cfg3.err; python tools/show_bench.py gpurun_out/r3l_cfg3.json > gpurun_out/r3l_cfg3.txt; head -26 gpurun_out/r3l_cfg3.txt | grep -v roofline
timeout -k 10 200 python tools/shard_step_bench.py 8 0 strong > gpurun_out/r3l_w8.txt 2>&1; tail -30 gpurun_out/r3l_w8.txt
