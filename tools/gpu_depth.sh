#!/bin/bash
# pipeline depth / stream grouping variants of the default command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for D in 3 4 5 4 5; do
  python bench.py --no-cpu-baseline --pipeline-depth $D 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $D:', d['value'], d['ms_per_step'])"
done
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rccl world 1:', d['value'], d['ms_per_step'], d['config'].get('batches_in_flight'))"
