mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3p_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3p_tests.log
timeout -k 10 120 python tools/cfg1_epoch_cpu.py > gpurun_out/r3p_cfg1_cpu_epoch.txt 2>&1; cat gpurun_out/r3p_cfg1_cpu_epoch.txt
timeout -k 10 600 python bench.py > gpurun_out/r3p_bench.json 2> gpurun_out/r3p_bench.err; echo "bench rc=$?"; python tools/show_bench.py gpurun_out/r3p_bench.json > gpurun_out/r3p_bench.txt; head -8 gpurun_out/r3p_bench.txt; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3p_bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
print('cpu', d.get('cpu_baseline'))
leg=d.get('cfg5shard_configs4'); print('leg', {k:leg.get(k) for k in ('value','ms_per_step','roofline','error')} if leg else None)
print('tpims', d.get('tpims_configs1',{}).get('value'))
PY
