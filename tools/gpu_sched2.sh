#!/bin/bash
# the stage schedule's knobs at the round's head: hardware queues, depth, selection lag, spare buffer set
mkdir -p gpurun_out/sched
o=gpurun_out/sched/sched4.txt; : > $o
run() { python bench.py --no-al-round --no-cpu-baseline "$@" 2>gpurun_out/sched/err.txt | python -c "import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print('%-30s Q=%-3s value %.1f ms/step %.3f' % (' '.join(sys.argv[1:]), os.environ.get('GPU_MAX_HW_QUEUES','-'), d['value'], d['ms_per_step']))" "$@" >> $o || tail -3 gpurun_out/sched/err.txt >> $o; }
for rep in 1 2; do for st in 20 100; do
run --steps $st --warmup 5
for q in 3 5 6 7; do GPU_MAX_HW_QUEUES=$q run --steps $st --warmup 5; done
done; done
cat $o
