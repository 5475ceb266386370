#!/bin/bash
# the N > 1 code path (RCCL exchanges on device buffers) on one GPU, next to the plain path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for D in 4 5; do
SSDR_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --precision bf16x3 --pipeline-depth $D 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('dist depth $D', d['value'], d['ms_per_step'])"
done
for D in 4 5; do
timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --precision bf16x3 --pipeline-depth $D 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('plain depth $D', d['value'], d['ms_per_step'])"
done
