#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pad in 8000 11000 14000; do
  SSDR_FE_PADLDS=$pad rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fepad/p$pad -o pf -- python3 tools/fe_bench.py 2 0 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/fepad/p$pad/*counter_collection.csv")[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "fe_reduce" in r["Kernel_Name"]]
print("pad $pad fe_reduce FETCH_SIZE x2 per launch MB:", round(2*sum(v)/len(v)*1024/1e6,1))
PY
done
for r in 1 2 3; do for pad in 0 8000 11000; do
  echo -n "bench pad $pad: "
  SSDR_FE_PADLDS=$pad timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['subsample+tile'], o['fe_reduce']['ms_per_step'])"
done; done
