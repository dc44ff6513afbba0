#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (clock, MFMA busy share, wait shares, HBM bytes):

    python tools/pmc_summary.py <dir> [<dir> ...] [top_n]

One line per kernel and directory.  FETCH_SIZE / WRITE_SIZE are in KB per launch as rocprofv3 reports them; on gfx950
FETCH_SIZE counts half the bytes of 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM) -- bench.py's ``traffic`` applies
that correction (2 x FETCH_SIZE + WRITE_SIZE) when it reads the summary tracked under profiles/."""
import collections
import csv
import glob
import sys

args = sys.argv[1:]
top = 8
if args and args[-1].isdigit():
    top = int(args.pop())
for d in args:
    files = glob.glob(d + "/*/*counter_collection.csv")
    if not files:
        print(f"# {d}: no counter_collection.csv")
        continue
    rows = list(csv.DictReader(open(files[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    dur = collections.defaultdict(float)
    seen = set()
    for r in rows:
        k = r["Kernel_Name"][:96]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:top]:
        n = cnt[k]
        dd = dur[k] / n
        line = f"{k:96s} n={n:3d} dur={dd / 1e3:8.1f}us"
        if "GRBM_GUI_ACTIVE" in v:
            clk = v["GRBM_GUI_ACTIVE"] / n / 8 / dd
            line += f" clk={clk:5.2f}GHz"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                line += f" mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (1024 * clk * dd):5.2f}"
        wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        for c, lab in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_ANY", "active")):
            if c in v:
                line += f" {lab}={v[c] / wc:5.2f}"
        for c in ("SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
            if c in v:
                line += f" {c}={v[c] / n:.3e}"
        print(line)
