#!/bin/bash
# the default command with more hardware queues than the runtime's default of 4 (every stream on a queue of its own from 8 on)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for Q in 4 8 16 8 4; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$Q:', d['value'], d['ms_per_step'])"
done
