#!/bin/bash
# the AL round leg alone, a few stream arrangements (GPU_MAX_HW_QUEUES as a cross-check)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for q in "" 8; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 600 python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['al_round']; print(d['value'], {k:a[k] for k in ('ms','Mpoints_per_s','inference_ms','selection_ms','fps_ms')})"
done
