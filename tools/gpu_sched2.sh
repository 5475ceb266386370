#!/bin/bash
# the stage schedule's knobs at the round's head: hardware queues, depth, selection lag, spare buffer set
mkdir -p gpurun_out/sched
o=gpurun_out/sched/sched3.txt; : > $o
run() { python bench.py --no-al-round --no-cpu-baseline "$@" 2>gpurun_out/sched/err.txt | python -c "import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print('%-60s Q=%-3s value %.1f ms/step %.3f' % (' '.join(sys.argv[1:]), os.environ.get('GPU_MAX_HW_QUEUES','-'), d['value'], d['ms_per_step']))" "$@" >> $o || tail -3 gpurun_out/sched/err.txt >> $o; }
for st in 20 300; do
run --steps $st --warmup 5
GPU_MAX_HW_QUEUES=8 run --steps $st --warmup 5
GPU_MAX_HW_QUEUES=12 run --steps $st --warmup 5
GPU_MAX_HW_QUEUES=2 run --steps $st --warmup 5
run --steps $st --warmup 5 --pipeline-depth 4
run --steps $st --warmup 5 --select-lag 2
run --steps $st --warmup 5 --spare-set
GPU_MAX_HW_QUEUES=8 run --steps $st --warmup 5 --select-lag 2
GPU_MAX_HW_QUEUES=8 run --steps $st --warmup 5 --spare-set
done
cat $o
