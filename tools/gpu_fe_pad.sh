#!/bin/bash
# fe_reduce with its resident workgroups per CU capped by unused LDS (does the walk's second read of a bucket's records hit L2 with fewer buckets in flight?)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for pad in 0 8000 14000 22000 35000; do
  echo -n "pad $pad: "; SSDR_FE_PADLDS=$pad timeout 120 python3 tools/fe_bench.py 30 0,0 2>&1 | tail -1
done; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pad in 0 22000; do
  SSDR_FE_PADLDS=$pad rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fepad/p$pad -o pf -- python3 tools/fe_bench.py 2 0 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/fepad/p$pad/*counter_collection.csv")[0]
acc={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "fe_reduce" in n: acc.setdefault("fe_reduce",[]).append(float(r["Counter_Value"]))
for k,v in acc.items(): print("pad $pad", k, "launches", len(v), "FETCH_SIZE x2 per launch MB:", round(2*sum(v)/len(v)*1024/1e6,1))
PY
done
