#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_al_round.py tests/test_select.py -x -q -m gpu > gpurun_out/r6a_tests.txt 2>&1; tail -15 gpurun_out/r6a_tests.txt
timeout 600 python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err; tail -3 gpurun_out/r6a_bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r6a_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms']); print(json.dumps(d['al_round'], indent=1))"
out=gpurun_out/fps6f.txt; : > $out
SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py 20000 10000 --save /tmp/seq.npy >> $out 2>&1
for dl in 0 2 4 6 8; do echo "delay $dl" >> $out; SSDR_FPS_DELAY=$dl timeout 300 python3 tools/fps_large.py 20000 10000 --cmp /tmp/seq.npy >> $out 2>&1; done
cat $out
