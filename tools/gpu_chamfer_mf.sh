#!/bin/bash
# same-box A/B of the chamfer kernels: float64 screening (SSDR_CHAMFER_F64=1) against the float32 screening on the matrix cores
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for f in 1 0; do
  echo -n "bench F64=$f: "
  SSDR_CHAMFER_F64=$f timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['select'], {k: o[k]['ms_per_step'] for k in o if k.startswith('sel_')})"
done; done
