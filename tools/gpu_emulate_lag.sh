#!/bin/bash
# the FPS load of N ranks (SSDR_EMULATE_WORLD) through the sharded code path, with 1 / 2 / 3 selections in flight behind the newest
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1
for N in 1 4 8; do for LAG in 1 2 3; do
  SSDR_EMULATE_WORLD=$N python3 bench.py --steps 100 --no-cpu-baseline --select-lag $LAG 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPS load of $N rank(s), lag $LAG:', d['value'], 'Mpoints/s per GPU,', d['ms_per_step'], 'ms per step')"
done; done
