#!/bin/bash
# A/B of the reduction's workgroup size on one box (abtmp/lib_nt128.so / lib_nt256.so), records read in place in both
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 WGS=$2: "; SSDR_AL_LIBRARY=$PWD/$1 SSDR_FE_WGS=$2 timeout 120 python3 tools/fe_bench.py 30 0,0 2>&1 | tail -1; }
for r in 1 2; do
  run abtmp/lib_nt256.so 8
  run abtmp/lib_nt128.so 8
  run abtmp/lib_nt128.so 16
  run abtmp/lib_nt128.so 12
done
for lib in abtmp/lib_nt256.so abtmp/lib_nt128.so abtmp/lib_nt256.so abtmp/lib_nt128.so; do
  echo -n "bench $lib: "
  SSDR_AL_LIBRARY=$PWD/$lib SSDR_FE_WGS=$([ $lib = abtmp/lib_nt128.so ] && echo 16 || echo 8) timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['subsample+tile'], d['roofline']['others']['fe_reduce']['ms_per_step'])"
done
