#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 3 4 5; do
  python bench.py --steps 80 --warmup 6 --no-cpu-baseline --pipeline-depth $D > gpurun_out/depth_$D.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/depth_$D.json')); print('depth $D:', d['value'], d['ms_per_step'])"
done
