#!/bin/bash
# pipelined bench under the kernel trace: which stage's kernels run on which hardware queue, and whether consecutive selection chains overlap
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-al-round > $OUT/bench.json 2> $OUT/kt.err
python - > $OUT/timeline.txt <<PY
import csv, json, collections
print(json.load(open("$OUT/bench.json"))["value"])
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0].split("<")[0]
fam = lambda n: "front" if n.startswith(("fe_","tile_")) else "knn" if n.startswith(("grid_","kd_")) else "net" if n.startswith(("lfa","dense","gather_max","tail_")) else "select" if n.startswith(("sel_chamfer","sel_adj","sel_prop","fps_","cand_","sel_segment","fill_double","sel_centres")) else "score" if n.startswith("sel_") else "other"
q=collections.defaultdict(lambda: collections.Counter())
for r in rows: q[(r["Queue_Id"], r.get("Stream_Id", "?"))][fam(nm(r))]+=1
for k,v in sorted(q.items()): print("queue %s stream %s: %s" % (k[0], k[1], dict(v)))
# steady-state window: selection chains
sel=[r for r in rows if nm(r) in ("fps_block_reg","sel_chamfer_dir_batch","sel_chamfer_plan","cand_rank")]
mid=sel[len(sel)//2:len(sel)//2+40]
t0=int(mid[0]["Start_Timestamp"])
for r in mid: print("%-24s queue %s stream %s  start %8.3f ms  end %8.3f ms  (%.3f)" % (nm(r), r["Queue_Id"], r.get("Stream_Id","?"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
rm -rf $OUT/kt
cat $OUT/timeline.txt
