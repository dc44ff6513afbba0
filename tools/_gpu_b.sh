set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "backward" > gpurun_out/r3t_tests.log 2>&1 || { tail -30 gpurun_out/r3t_tests.log; exit 1; }
tail -2 gpurun_out/r3t_tests.log
timeout -k 10 200 python tools/fused_trace.py bwd > gpurun_out/r3t_trace_bwd.txt 2>&1; tail -8 gpurun_out/r3t_trace_bwd.txt
timeout -k 10 300 python bench.py --workload cfg5shard --no-cfg5-leg --no-split-leg --no-tpims-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3t_cfg5.json 2> gpurun_out/r3t.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3t_cfg5.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:round(v['avg_ms']*v['launches']/20,3) for k,v in d['stages'].items() if v['avg_ms']*v['launches']/20>0.3})
PY
