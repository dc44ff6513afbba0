mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_bf16.py tests/test_gpu_switches.py tests/test_gpu_fullsize_cfg5.py -x -q -m gpu > gpurun_out/r3w_tests.log 2>&1; tail -3 gpurun_out/r3w_tests.log
for dbg in 0; do
REGT_FUSED_DBG=$dbg timeout -k 10 300 python bench.py --workload cfg5shard --no-cfg5-leg --no-split-leg --no-tpims-leg --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r3x_$dbg.json 2> gpurun_out/r3x.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r3x_$dbg.json').read().strip().splitlines()[-1])
print('dbg=$dbg', round(d['ms_per_step'],3), 'fused_forward', round(d['stages']['fused_forward']['avg_ms'],3))
PY
done
