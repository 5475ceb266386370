#!/bin/bash
# the AL round leg alone: batches in flight (SSDR_AL_SLOTS) x hardware queues (GPU_MAX_HW_QUEUES)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/al_slots.txt; : > $out
for q in 4 6 8; do for sl in 4 6 8; do
  echo -n "GPU_MAX_HW_QUEUES=$q SSDR_AL_SLOTS=$sl: " >> $out
  GPU_MAX_HW_QUEUES=$q SSDR_AL_SLOTS=$sl timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['al_round']; print({k:a[k] for k in ('ms','Mpoints_per_s','inference_ms','selection_ms','fps_ms')})" >> $out
done; done
cat $out
