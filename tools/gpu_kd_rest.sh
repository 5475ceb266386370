#!/bin/bash
# tree hand-over: level-wide launches down to the balanced depth + margin, then one launch for what is still open (SSDR_KD_REST_MARGIN; 99 = level-wide all the way), same box
cd $GRAFT_REPO_ROOT
for r in 1 2; do for m in 99 0 1 2 3 5; do
  echo -n "margin $m: "
  SSDR_KD_REST_MARGIN=$m timeout 120 python3 bench.py --no-cpu-baseline --steps 100 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['knn_pyramid'], o['knn_tree_handover']['ms_per_step'])"
done; done
