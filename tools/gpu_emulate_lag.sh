#!/bin/bash
# emulated N = 4 / 8 ranks (FPS load) against the number of selection chains kept in flight (--select-lag) and the hardware queues
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1
for N in 4 8; do for lag in 1 2 3 4; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q SSDR_EMULATE_WORLD=$N python3 bench.py --steps 60 --no-cpu-baseline --select-lag $lag 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$N lag=$lag hwq=$q:', d['value'], 'Mpoints/s,', d['ms_per_step'], 'ms/step')"
done; done; done
