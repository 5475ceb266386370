#!/bin/bash
# the sharded code path (RCCL, world size 1) with the runtime's default 4 hardware queues and with 8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for Q in 4 8 4 8; do
  GPU_MAX_HW_QUEUES=$Q MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rccl world 1, GPU_MAX_HW_QUEUES=$Q:', d['value'], d['ms_per_step'])"
done
