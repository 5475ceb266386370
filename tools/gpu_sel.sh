#!/bin/bash
# selection GPU check: parity tests, bench line, per-kernel stats of the sel_* / fps kernels (sequential)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_select.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_stats.sh "sel_ fps_" bf16x3
python bench.py --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/stats/bench_pipe.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/stats/bench_pipe.json')); print('pipelined:', d['value'], d['ms_per_step'])"
