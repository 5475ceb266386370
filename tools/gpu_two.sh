#!/bin/bash
# two independent bench processes on the one GPU: is there throughput left that one pipeline does not reach?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 150 --warmup 10 --no-cpu-baseline > gpurun_out/two_a.json 2>/dev/null &
P1=$!
python bench.py --steps 150 --warmup 10 --no-cpu-baseline > gpurun_out/two_b.json 2>/dev/null &
P2=$!
wait $P1; wait $P2
python -c "
import json
a=json.load(open('gpurun_out/two_a.json')); b=json.load(open('gpurun_out/two_b.json'))
print('two processes:', a['value'], '+', b['value'], '=', a['value']+b['value'])"
