#!/bin/bash
# same-box A/B of two builds, network families shown: tools/gpu_ab_net.sh abtmp/lib_a.so abtmp/lib_b.so
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in "$@"; do
  echo -n "bench $lib: "
  SSDR_AL_LIBRARY=$PWD/$lib timeout 120 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['randla_infer'], d['roofline']['frac'], {k: o[k]['ms_per_step'] for k in ('lfa_att_kernel','dense_kernel','tail_kernel','gather_max_kernel')})"
done; done
