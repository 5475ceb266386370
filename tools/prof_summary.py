#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats csv directory: per-kernel calls, average us, ms per bench step."""
import csv, glob, sys
d = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 11.0     # 2 warmup + 5 timed + 3 prof + 1 staged
rows = list(csv.DictReader(open(glob.glob(d + '/*kernel_stats.csv')[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("kernel busy ms/step: %.3f" % (tot / 1e6 / steps))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = r['Name'].replace('ssdr::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    print("%-58s %6s %9.1f us avg %7.3f ms/step" % (n[:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6 / steps))
