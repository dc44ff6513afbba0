mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3v_tests.log 2>&1; tail -3 gpurun_out/r3v_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r3v_bench.json 2> gpurun_out/r3v_bench.err; tail -c 3000 gpurun_out/r3v_bench.json
