#!/usr/bin/env python3
"""Per-kernel table from two rocprofv3 --pmc passes (tools/gpu_pmc.sh): pmc_table.py <dir> [name filter ...]"""
import collections, csv, sys
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("ssdr::", "").split("(")[0]
def load(f):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        d[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n].add(r["Dispatch_Id"])
        d[n]["vgpr"] = float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]); d[n]["scr"] = float(r["Scratch_Size"])
    return d, cnt
d1, c1 = load(sys.argv[1] + "/pass1.csv"); d2, c2 = load(sys.argv[1] + "/pass2.csv")
flt = sys.argv[2:]
for n in sorted(d1):
    if flt and not any(f in n for f in flt): continue
    a = d1[n]; b = d2.get(n, {}); k = len(c1[n]); wc = max(a["SQ_WAVE_CYCLES"], 1); W = max(a["SQ_WAVES"], 1)
    print("%-36s disp %3d waves/disp %7d regs %3d scr %4d | cyc/wave %8.0f wait_any %3.0f%% wait_inst %3.0f%% active %3.0f%% | per wave: valu %7.0f vmem %6.0f salu %6.0f lds %5.0f branch %6.0f | mfma busy %4.1f%%" % (
        n[:36], k, W / k, a["vgpr"], a["scr"], wc * 4 / W, 100 * a["SQ_WAIT_ANY"] / wc, 100 * a["SQ_WAIT_INST_ANY"] / wc, 100 * a["SQ_ACTIVE_INST_ANY"] / wc,
        a["SQ_INSTS_VALU"] / W, a["SQ_INSTS_VMEM"] / W, b.get("SQ_INSTS_SALU", 0) / W, b.get("SQ_INSTS_LDS", 0) / W, b.get("SQ_INSTS_BRANCH", 0) / W,
        100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(a.get("SQ_BUSY_CYCLES", 1), 1)))
