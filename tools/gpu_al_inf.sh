#!/bin/bash
# kernel trace of the AL round's inference half (second run of tools/al_sel_probe.py): busy time of every hardware queue, idle time of the whole GPU, the kernels by total time
OUT=$GRAFT_REPO_ROOT/gpurun_out/alinf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 tools/al_sel_probe.py fps > $OUT/run.txt 2> $OUT/kt.err
python - > $OUT/alinf.txt <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
# the second round: from the first fe_minmax_partial after the first fps_coop_split to the second fps_coop_split
fps=[i for i,r in enumerate(rows) if "fps_coop_split" in r["Kernel_Name"]]
a=next(i for i in range(fps[-2]+1,len(rows)) if "fe_minmax_partial" in rows[i]["Kernel_Name"])
b=max(i for i,r in enumerate(rows) if i<fps[-1] and ("tail_bf16" in r["Kernel_Name"]))
seg=rows[a:b+1]
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
print("inference half: %d dispatches, %.2f ms" % (len(seg), (t1-t0)/1e6))
iv=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in seg)
busy=0; cs,ce=iv[0]
gaps=[]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; gaps.append((s-ce, (ce-t0)/1e6)); cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("GPU busy (union of kernels) %.2f ms = %.1f %%; idle gaps: %d, the largest (us @ ms): %s" % (busy/1e6, 100*busy/(t1-t0), len(gaps), [(round(g/1e3,1), round(t,2)) for g,t in sorted(gaps,reverse=True)[:8]]))
q=collections.defaultdict(float)
for r in seg: q[(r["Queue_Id"], r.get("Stream_Id","?"))]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in sorted(q.items()): print("queue %s stream %s busy %.2f ms (%.0f %%)" % (k[0], k[1], v/1e6, 100*v/(t1-t0)))
# concurrency: time-weighted number of kernels in flight
ev=[]
for s,e in iv: ev.append((s,1)); ev.append((e,-1))
ev.sort(); cur=0; last=t0; hist=collections.defaultdict(int)
for t,d in ev: hist[cur]+=t-last; last=t; cur+=d
print("kernels in flight: " + ", ".join("%d: %.1f %%" % (k, 100*v/(t1-t0)) for k,v in sorted(hist.items())))
k=collections.defaultdict(float); c=collections.Counter()
for r in seg: k[nm(r).split("<")[0]]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); c[nm(r).split("<")[0]]+=1
for n,v in sorted(k.items(), key=lambda x:-x[1])[:24]: print("  %-34s %5d x  %8.3f ms total  (%.3f per batch)" % (n[:34], c[n], v/1e6, v/1e6/17))
print("sum of kernel time %.2f ms (%.2f per batch)" % (sum(k.values())/1e6, sum(k.values())/1e6/17))
PY
rm -rf $OUT/kt
cat $OUT/alinf.txt
