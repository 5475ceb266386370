#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for T in 14 17 19 22 26 30; do
  SSDR_KNN_TARGET_PTS=$T python bench.py --steps 40 --warmup 4 --no-cpu-baseline --stages > gpurun_out/tp_$T.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/tp_$T.json')); print('target', $T, d['value'], d['ms_per_step'], d['stage_ms']['knn_pyramid'], {k:v['ms_per_step'] for k,v in d['roofline']['others'].items() if 'knn' in k})"
done
