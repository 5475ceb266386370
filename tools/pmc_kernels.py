#!/usr/bin/env python3
"""Per-kernel sums of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_kernels.py <dir> [name filter]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("ssdr::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and flt not in n:
        continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(n, r["Counter_Name"])] += 1
for n in acc:
    print(n)
    for c, v in sorted(acc[n].items()):
        k = calls[(n, c)]
        print("    %-28s %16.0f per launch (%d launches)" % (c, v / k, k))
