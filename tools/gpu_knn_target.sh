#!/bin/bash
# sweep of the grid's cell-size target (points around a sample that fix the cell size)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for tp in 0 14 18 26 32; do
  echo -n "target $tp: "
  SSDR_KNN_TARGET_PTS=$tp timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['knn_pyramid'], {k: o[k]['ms_per_step'] for k in o if k.startswith('knn')})"
done; done
