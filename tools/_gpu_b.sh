mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3o_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3o_tests.log
timeout -k 10 200 python bench.py --workload cfg5shard --no-cpu-baseline --no-split-leg --steps 20 > gpurun_out/r3o_cfg5.json 2> gpurun_out/r3o_cfg5.err; python tools/show_bench.py gpurun_out/r3o_cfg5.json | head -3
