#!/bin/bash
# default bench under the kernel trace: how often do the FPS kernels (one workgroup, 1.4 ms) of consecutive batches run at the same time, per-queue busy time
OUT=$GRAFT_REPO_ROOT/gpurun_out/fpsov
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/kt.err
python3 - <<PY
import csv, json, collections
print(json.load(open("$OUT/bench.json"))["value"])
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
f=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id","")) for r in rows if "fps_block" in r["Kernel_Name"]]
f.sort()
f=f[10:-5]
ov=0; tot=0
for i in range(len(f)-1):
    s,e,_,_=f[i]; s2,e2,_,_=f[i+1]
    tot+=e-s; ov+=max(0,min(e,e2)-s2)
print("FPS kernels:", len(f), "mean duration %.3f ms" % (tot/len(f)/1e6), "overlap with the next one: %.1f%% of its duration" % (100*ov/tot))
print("gap start-to-start %.3f ms" % ((f[-1][0]-f[0][0])/(len(f)-1)/1e6))
print("queues of consecutive FPS kernels:", [x[2] for x in f[:8]])
q=collections.defaultdict(set)
for r in rows:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0][:28]
    q[r["Queue_Id"]].add(n)
for k,v in q.items(): print("queue", k, sorted(v)[:14])
PY
rm -rf $OUT/kt
