#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_pipeline.py tests/test_distributed.py tests/test_knn.py -m gpu -x -q 2>&1 | tail -3
for Q in default 8; do
  if [ $Q != default ]; then export GPU_MAX_HW_QUEUES=$Q; fi
  python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/ov_$Q.json 2>gpurun_out/ov_$Q.err
  python -c "import json; d=json.load(open('gpurun_out/ov_$Q.json')); print('queues $Q:', d['value'], d['ms_per_step'])"
done
unset GPU_MAX_HW_QUEUES
python tools/hosttime.py 2>&1 | tail -9
