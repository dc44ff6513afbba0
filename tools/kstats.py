#!/usr/bin/env python3
"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats CSV: kstats.py <dir> <steps>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (us): %.1f" % (tot / steps / 1e3))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print("%-62s calls/step=%5.1f avg_us=%8.1f us/step=%8.1f" % (r["Name"][:62], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                               float(r["TotalDurationNs"]) / steps / 1e3))
