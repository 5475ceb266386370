#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do for sl in 16 8 4 3 2 1; do
  echo -n "slices $sl: "
  SSDR_CHAMFER_SLICES=$sl timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['select'], o['sel_chamfer']['ms_per_step'])"
done; done
