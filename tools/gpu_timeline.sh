#!/bin/bash
# pipelined bench under the kernel trace: GPU busy fraction (union of kernel intervals), per-queue busy time, top kernels
P=${1:-bf16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --precision $P > $OUT/bench.json 2> $OUT/kt.err
python - <<PY
import csv, json, collections
print(json.load(open("$OUT/bench.json"))["value"])
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# timed region = the 30 steps: take the middle 60% of the trace by time
t0=int(rows[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in rows)
# find the steady-state window: from 40% to 80% of kernels
a=int(rows[int(len(rows)*0.35)]["Start_Timestamp"]); b=int(rows[int(len(rows)*0.75)]["Start_Timestamp"])
iv=[(max(int(r["Start_Timestamp"]),a),min(int(r["End_Timestamp"]),b),r) for r in rows if int(r["End_Timestamp"])>a and int(r["Start_Timestamp"])<b]
iv.sort(key=lambda x:x[0])
busy=0; cur_s=None; cur_e=None
for s,e,_ in iv:
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("window %.1f ms, GPU busy (union) %.1f%%" % ((b-a)/1e6, 100*busy/(b-a)))
q=collections.defaultdict(float); k=collections.defaultdict(float)
for s,e,r in iv:
    q[r["Queue_Id"]]+=e-s
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
    k[n]+=e-s
nfps=sum(1 for s,e,r in iv if "fps_block" in r["Kernel_Name"])
print("steps in window ~", nfps, " ms/step ~ %.2f" % ((b-a)/1e6/max(nfps,1)))
for qq,v in sorted(q.items()): print("queue", qq, "busy %.1f%%" % (100*v/(b-a)))
for n,v in sorted(k.items(), key=lambda x:-x[1])[:28]: print("  %-40s %.3f ms/step" % (n[:40], v/1e6/max(nfps,1)))
print("sum of kernel time per step: %.2f ms" % (sum(k.values())/1e6/max(nfps,1)))
PY
rm -rf $OUT/kt
