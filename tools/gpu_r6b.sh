#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python3 bench.py --steps 100 --no-cpu-baseline > gpurun_out/r6b_bench.json 2> gpurun_out/r6b_bench.err; tail -3 gpurun_out/r6b_bench.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r6b_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms']); print(json.dumps(d['al_round'], indent=1)); r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us']); print(json.dumps({k:v for k,v in r['others'].items()}, indent=0)[:6000])"
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r6b_tests.txt 2>&1; tail -5 gpurun_out/r6b_tests.txt
