#!/bin/bash
# the N > 1 code path (RCCL exchanges, torch's streams in the process) at world size 1: pipeline depth and queue count
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/dist1.txt; : > $out
run() { MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SSDR_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-al-round "$@" 2>gpurun_out/dist1.err | python3 -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s Q=%-3s value %.1f ms/step %.3f' % (' '.join(sys.argv[1:]), os.environ.get('GPU_MAX_HW_QUEUES','-'), d['value'], d['ms_per_step']))" "$@" >> $out || tail -2 gpurun_out/dist1.err >> $out; }
for st in 20 100; do
run --steps $st --warmup 5
run --steps $st --warmup 5 --pipeline-depth 5
run --steps $st --warmup 5 --pipeline-depth 3
SSDR_NCCL_NORMAL_PRIO=1 run --steps $st --warmup 5
SSDR_NCCL_NORMAL_PRIO=1 run --steps $st --warmup 5 --pipeline-depth 5
GPU_MAX_HW_QUEUES=5 run --steps $st --warmup 5
GPU_MAX_HW_QUEUES=5 run --steps $st --warmup 5 --pipeline-depth 5
GPU_MAX_HW_QUEUES=6 run --steps $st --warmup 5 --pipeline-depth 5
done
cat $out
