#!/bin/bash
# round 6: members in flight in fe_reduce's ordered walk (abtmp/lib_walk{4,6,8}.so), alternated on one box
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for w in 4 6 8; do
  echo -n "walk $w: "; SSDR_AL_LIBRARY=$PWD/abtmp/lib_walk$w.so timeout 120 python3 tools/fe_bench.py 40 0 2>&1 | tail -1
done; done
for w in 4 8 4 8; do
  echo -n "bench walk $w: "
  SSDR_AL_LIBRARY=$PWD/abtmp/lib_walk$w.so timeout 120 python3 bench.py --no-cpu-baseline --no-al-round --steps 200 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms']['subsample+tile'], d['roofline']['others']['fe_reduce']['ms_per_step'])"
done
