#!/bin/bash
# kernel trace of the AL round's selection half: every dispatch behind the last network kernel, in order, with the idle time in front of it
OUT=$GRAFT_REPO_ROOT/gpurun_out/alsel
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 tools/al_sel_probe.py ${1:-fps} > $OUT/run.txt 2> $OUT/kt.err
python - > $OUT/alsel_${1:-fps}.txt <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
last=max(i for i,r in enumerate(rows) if "tail_bf16" in r["Kernel_Name"] or "gather_max" in r["Kernel_Name"] or "lfa32" in r["Kernel_Name"])
sel=rows[last+1:]
t0=int(rows[last]["End_Timestamp"]); prev=t0; tot=0
print("selection half: %d dispatches, %.3f ms from the end of the last network kernel to the end of the last kernel" % (len(sel), (int(sel[-1]["End_Timestamp"])-t0)/1e6))
for r in sel:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-46s start %9.3f ms  gap %7.1f us  dur %9.1f us  grid %s" % (nm(r)[:46], (s-t0)/1e6, (s-prev)/1e3, (e-s)/1e3, r.get("Grid_Size_X", r.get("Grid_Size",""))))
    prev=max(prev,e); tot+=e-s
print("sum of durations %.3f ms" % (tot/1e6))
PY
rm -rf $OUT/kt
cat $OUT/alsel_${1:-fps}.txt
