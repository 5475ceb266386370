#!/bin/bash
# which hardware queue each stream lands on, for several creation orders / queue counts (rocprofv3 kernel trace, 6 steps each)
OUT=$GRAFT_REPO_ROOT/gpurun_out/qdisc
mkdir -p $OUT; : > $OUT/map.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() {
  rm -rf $OUT/kt
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-al-round > $OUT/bench.json 2> $OUT/kt.err
  python - >> $OUT/map.txt <<PY
import csv, json, collections, os
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0].split("<")[0]
fam = lambda n: "front" if n.startswith(("fe_","tile_")) else "knn" if n.startswith(("grid_","kd_")) else "net" if n.startswith(("lfa","dense","gather_max","tail_")) else "select" if n.startswith(("sel_chamfer","sel_adj","sel_prop","fps_","cand_","sel_segment","fill_double","sel_centres")) else "score" if n.startswith("sel_") else "other"
q=collections.defaultdict(lambda: collections.Counter())
for r in rows: q[(r["Queue_Id"], r.get("Stream_Id", "?"))][fam(nm(r))]+=1
print("QMAP=%s HWQ=%s value %s" % (os.environ.get("SSDR_PIPE_QMAP","default"), os.environ.get("GPU_MAX_HW_QUEUES","-"), json.load(open("$OUT/bench.json"))["value"]))
for k,v in sorted(q.items()):
    top = max(v, key=v.get)
    print("   queue %s stream %s: %s (%d kernels)" % (k[0], k[1], top if v[top] > 0.6*sum(v.values()) else dict(v), sum(v.values())))
PY
}
one
GPU_MAX_HW_QUEUES=5 one
GPU_MAX_HW_QUEUES=8 one
SSDR_PIPE_QMAP=2,3,0,1,1,1 one
SSDR_PIPE_QMAP=0,1,2,3,-,3 one
SSDR_PIPE_QMAP=2,0,3,1,-,3 one
rm -rf $OUT/kt
cat $OUT/map.txt
