#!/bin/bash
# the AL round's kd_tree_kernel calls (the hand-over's one-workgroup-per-tree build): when and how long, second round of tools/al_sel_probe.py
OUT=$GRAFT_REPO_ROOT/gpurun_out/alkd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SSDR_KNN_DEBUG=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 tools/al_sel_probe.py fps > $OUT/run.txt 2> $OUT/kt.err
python - > $OUT/alkd.txt <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
fps=[i for i,r in enumerate(rows) if "fps_coop_split" in r["Kernel_Name"]]
a=next(i for i in range(fps[-2]+1,len(rows)) if "fe_minmax_partial" in rows[i]["Kernel_Name"])
t0=int(rows[a]["Start_Timestamp"])
tend=max(int(r["End_Timestamp"]) for r in rows[a:fps[-1]] if "tail_bf16" in r["Kernel_Name"])
print("inference half of the second round: %.2f ms" % ((tend-t0)/1e6))
for r in rows[a:fps[-1]]:
    n=r["Kernel_Name"]
    if "kd_tree_kernel" in n or "kd_search_worklist_kernel<16" in n:
        print("%-28s stream %s start %7.2f ms dur %8.1f us" % (n.split("(")[0].replace("ssdr::(anonymous namespace)::","").replace("void ","")[:28], r.get("Stream_Id","?"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
rm -rf $OUT/kt
cat $OUT/alkd.txt; grep -i "hand\|tree\|ball" $OUT/kt.err | head -40
