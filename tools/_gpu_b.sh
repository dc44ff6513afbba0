mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3y_tests.log 2>&1; tail -3 gpurun_out/r3y_tests.log
timeout -k 10 300 python bench.py --workload cfg3 --no-cfg5-leg --no-split-leg --no-tpims-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3y_cfg3.json 2> gpurun_out/r3y.err
python tools/show_bench.py gpurun_out/r3y_cfg3.json 2>&1 | sed -n 1,12p
