#!/bin/bash
# A/B of the front end's reduction on one box: LDS record image (3 workgroups per CU) against records read in place (4 / 5 / 6 waves per SIMD builds in abtmp/).
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 IMAGE=$2 WGS=$3: "; SSDR_AL_LIBRARY=$PWD/$1 SSDR_FE_IMAGE=$2 SSDR_FE_WGS=$3 timeout 120 python3 tools/fe_bench.py 30 0,0 2>&1 | tail -1; }
for r in 1 2; do
  run abtmp/lib_w4.so 1 0
  run abtmp/lib_w4.so 0 4
  run abtmp/lib_w4.so 0 8
  run abtmp/lib_w5.so 0 5
  run abtmp/lib_w5.so 0 10
  run abtmp/lib_w6.so 0 6
  run abtmp/lib_w6.so 0 12
done
for lib in "abtmp/lib_w4.so 1" "abtmp/lib_w4.so 0" "abtmp/lib_w5.so 0" "abtmp/lib_w6.so 0" "abtmp/lib_w4.so 1" "abtmp/lib_w4.so 0" "abtmp/lib_w5.so 0" "abtmp/lib_w6.so 0"; do
  set -- $lib
  echo -n "bench $1 IMAGE=$2: "
  SSDR_AL_LIBRARY=$PWD/$1 SSDR_FE_IMAGE=$2 timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['subsample+tile'], d['roofline']['others']['fe_reduce']['ms_per_step'])"
done
