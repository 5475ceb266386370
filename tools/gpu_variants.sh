#!/bin/bash
# bench.py in its other configurations: each must print one JSON line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo "== $*"; python bench.py "$@" --no-cpu-baseline 2>gpurun_out/var.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): d = json.loads(l); print('  ', d['value'], d['ms_per_step'], d['dtype'][:12], d['config'].get('batches_in_flight'))" || tail -3 gpurun_out/var.err; }
run --steps 20 --warmup 2 --selector kcenter
run --steps 20 --warmup 2 --precision f32
run --steps 20 --warmup 2 --precision bf16
run --steps 5 --warmup 1 --no-pipeline
run --steps 3 --warmup 0
run --steps 1 --warmup 1
run --gpus 1 --steps 10 --warmup 2 --pipeline-depth 2
