mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r3n_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3n_tests.log
timeout -k 10 200 python tools/fused_trace.py 2>&1 | tail -10
timeout -k 10 200 python bench.py --workload cfg5shard --no-cpu-baseline --no-split-leg --steps 20 > gpurun_out/r3n_cfg5.json 2> gpurun_out/r3n_cfg5.err; python tools/show_bench.py gpurun_out/r3n_cfg5.json > gpurun_out/r3n_cfg5.txt; head -24 gpurun_out/r3n_cfg5.txt
