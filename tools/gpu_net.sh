#!/bin/bash
# network-only GPU check: parity tests + sequential per-kernel stats for one precision
P=${1:-bf16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/net
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_randla.py -m gpu -x -q -s 2>&1 | tail -8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --precision $P > $OUT/bench_seq.json 2> $OUT/kts.err
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kts/kts_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "lfa" in n or "dense" in n or "tail" in n or "gather_max" in n:
        short=n.split("(")[0].replace("void ssdr::","").replace("ssdr::","")
        per=float(r["TotalDurationNs"])/int(r["Calls"])*(int(r["Calls"])/19.0)/1e3
        tot+=per
        print("  %-46s calls/step %5.1f  avg %8.1f us  per-step %8.1f us"%(short, int(r["Calls"])/19.0, float(r["AverageNs"])/1e3, per))
print("  network kernels per step: %.1f us" % tot)
PY
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kts/kts_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last full step: dispatches between the last two fc0 launches (dense_rows_kernel<6, 0, 8>)
idx=[i for i,r in enumerate(rows) if "dense_rows_kernel<6" in r["Kernel_Name"]]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n=r["Kernel_Name"].split("(")[0].replace("void ssdr::","").replace("ssdr::","").replace("(anonymous namespace)::","")
    if "kd_" in n or "rs_" in n or "gs_" in n or "sel_" in n or "tile_" in n or "fps" in n: continue
    print("%9.1f us  +%7.1f  %-40s grid %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, n, [r.get(k) for k in ("Grid_Size_X","Grid_Size_Y","Workgroup_Size_X","LDS_Block_Size","VGPR_Count","Accum_VGPR_Count")]))
PY
rm -rf $OUT/kts
