#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (stage table sorted by total time)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value={d['value']:.2f} {d['unit']}  ms/step={d['ms_per_step']:.3f}  n_gpus={d['n_gpus']}")
for k in ("roofline", "roofline_spmm", "mfma_all_gemms", "opt_in_bf16x3_split", "tpims_configs1", "cpu_baseline"):
    if k in d:
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d[k].items() if a not in ("sample", "note", "workload")})
if "stages" in d:
    tot = 0.0
    for k, v in sorted(d["stages"].items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
        per_step = v["avg_ms"] * v["launches"] / d["steps"]
        tot += per_step
        print(f"  {k:18s} launches={v['launches']:4d} avg_ms={v['avg_ms']:8.3f} ms/step={per_step:8.3f}")
    print(f"  sum of stages per step = {tot:.3f} ms")
