#!/bin/bash
# deliberate groupings of the six streams over the hardware queues (SSDR_PIPE_QMAP = front,knn,infer,score,selA,selB -> queue id; "-" = the library stream on queue 1;
# the NULL stream idles on queue 2), same box; HWQ = GPU_MAX_HW_QUEUES
mkdir -p gpurun_out/sched
o=gpurun_out/sched/qmap3.txt; : > $o
run() { python bench.py --no-al-round --no-cpu-baseline "$@" 2>gpurun_out/sched/err.txt | python -c "import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print('QMAP=%-14s HWQ=%s %-10s value %.1f ms/step %.3f' % (os.environ.get('SSDR_PIPE_QMAP','default'), os.environ.get('GPU_MAX_HW_QUEUES','4'), ' '.join(sys.argv[1:3]), d['value'], d['ms_per_step']))" "$@" >> $o || tail -3 gpurun_out/sched/err.txt >> $o; }
for st in 20 100; do
run --steps $st --warmup 5
for m in 4,4,3,2,-,1 4,4,3,3,-,2 4,4,3,1,-,2 4,4,3,2,-,2 4,4,3,3,-,1 4,3,2,4,-,1 4,3,2,2,-,1 4,3,2,1,-,1 3,3,4,4,-,2 4,4,2,2,-,3; do
SSDR_PIPE_QMAP=$m run --steps $st --warmup 5
done
for m in 4,5,3,3,-,2 4,4,3,5,-,2 4,4,3,3,-,5 4,5,3,2,-,1; do
GPU_MAX_HW_QUEUES=5 SSDR_PIPE_QMAP=$m run --steps $st --warmup 5
done
done
cat $o
