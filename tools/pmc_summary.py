#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (clock, MFMA busy share, wait shares)."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in rows:
    k = r["Kernel_Name"][:58]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        cnt[k] += 1
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    n = cnt[k]
    d = dur[k] / n
    line = f"{k:58s} n={n:3d} dur={d / 1e3:8.1f}us"
    if "GRBM_GUI_ACTIVE" in v:
        clk = v["GRBM_GUI_ACTIVE"] / n / 8 / d
        line += f" clk={clk:5.2f}GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            line += f" mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (1024 * clk * d):5.2f}"
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
        if c in v:
            line += f" {lab}={v[c] / wc:5.2f}"
    for c in ("SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
        if c in v:
            line += f" {c}={v[c] / n:.3e}"
    print(line)
