#!/bin/bash
# which hardware queue each stage / selection stream lands on (SSDR_PIPE_QMAP = front,knn,infer,score,selA,selB; "-" = the library stream), same box
mkdir -p gpurun_out/sched
o=gpurun_out/sched/qmap2.txt; : > $o
run() { python bench.py --no-al-round --no-cpu-baseline "$@" 2>gpurun_out/sched/err.txt | python -c "import sys,json,os; d=json.loads(sys.stdin.readlines()[-1]); print('QMAP=%-14s %-22s value %.1f ms/step %.3f' % (os.environ.get('SSDR_PIPE_QMAP','default'), ' '.join(sys.argv[1:3]), d['value'], d['ms_per_step']))" "$@" >> $o || tail -3 gpurun_out/sched/err.txt >> $o; }
run --steps 100 --warmup 5
for p in $(python -c "
import itertools
print(' '.join(','.join(map(str,p)) for p in itertools.permutations(range(4))))"); do
SSDR_PIPE_QMAP=$p,-,3 run --steps 100 --warmup 5
done
run --steps 100 --warmup 5
for b in 0 1 2; do SSDR_PIPE_QMAP=3,0,1,2,-,$b run --steps 100 --warmup 5; done
for a in 0 1 2 3; do SSDR_PIPE_QMAP=3,0,1,2,$a,3 run --steps 100 --warmup 5; done
cat $o
