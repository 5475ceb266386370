#!/bin/bash
# stream per STAGE (pipeline.Pipelined) against stream per BATCH (pipeline.BatchStreams) on one box: the bench's value at the driver's 20 steps and at 300
mkdir -p gpurun_out/sched
o=gpurun_out/sched/sched2.txt; : > $o
run() { python bench.py --no-al-round --no-cpu-baseline "$@" 2>gpurun_out/sched/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-70s value %.1f ms/step %.3f' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']))" "$@" >> $o || tail -3 gpurun_out/sched/err.txt >> $o; }
for st in 20 300; do
run --steps $st --warmup 5
for sl in 3 4 5 6; do for ss in 1 2 3; do
run --steps $st --warmup 5 --schedule batch --slots $sl --sel-streams $ss
done; done
done
cat $o
