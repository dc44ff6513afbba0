mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_bf16.py tests/test_gpu_nccl_single.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r3z_tests.log 2>&1; tail -3 gpurun_out/r3z_tests.log
timeout -k 10 300 python bench.py --workload cfg5shard --no-cfg5-leg --no-split-leg --no-tpims-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3z_cfg5.json 2> gpurun_out/r3z.err
python tools/show_bench.py gpurun_out/r3z_cfg5.json 2>&1 | sed -n 1,16p
