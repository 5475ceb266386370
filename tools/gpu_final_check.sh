#!/bin/bash
# what the driver runs at the end of a round, timed: the -m gpu suite, smoke(), the default bench
mkdir -p gpurun_out/final
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/final/gpu_tests.txt 2>&1
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/final/smoke.txt 2>&1
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/final/bench20.json 2> gpurun_out/final/bench20.err
( time python bench.py ) > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
tail -4 gpurun_out/final/gpu_tests.txt; tail -4 gpurun_out/final/smoke.txt; tail -4 gpurun_out/final/bench20.err; tail -4 gpurun_out/final/bench_default.err
python - <<'PY'
import json
for f in ("bench20", "bench_default"):
    d = json.loads(open("gpurun_out/final/%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["steps"], d["al_round"].get("ms"), d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
