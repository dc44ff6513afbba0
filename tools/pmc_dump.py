#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 --pmc output directories.

    python tools/pmc_dump.py <kernel-name substring> <dir> [<dir> ...]
"""
import collections
import csv
import glob
import sys

pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(float)
        disp = set()
        dur = 0.0
        for r in csv.DictReader(open(f)):
            if pat not in r["Kernel_Name"]:
                continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp:
                disp.add(r["Dispatch_Id"])
                dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n = max(len(disp), 1)
        print(f"{d}: {len(disp)} dispatches of *{pat}*, avg {dur / n / 1e3:.1f} us")
        for k, v in sorted(agg.items()):
            print(f"    {k:28s} {v / n:14.4e}")
