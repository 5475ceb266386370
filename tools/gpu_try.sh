#!/bin/bash
# quick GPU round trip: network parity in the three arithmetic modes, bench lines, per-kernel stats (sequential) per mode
OUT=$GRAFT_REPO_ROOT/gpurun_out/try
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_randla.py -m gpu -x -q -s 2>&1 | tail -12 > $OUT/test_randla.log
for P in f32 bf16x3 bf16; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --stages --precision $P > $OUT/bench_$P.json 2> $OUT/bench_$P.err
done
for P in f32 bf16x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts_$P -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --precision $P > /dev/null 2> $OUT/kts_$P.err
  cp $OUT/kts_$P/kts_kernel_stats.csv $OUT/seq_kernel_stats_$P.csv; rm -rf $OUT/kts_$P
done
cat $OUT/test_randla.log
for P in f32 bf16x3 bf16; do python - <<PY
import json
d=json.load(open("$OUT/bench_$P.json")); print("$P", d["value"], d["ms_per_step"], d["stage_ms"], {k:v["ms_per_step"] for k,v in d["roofline"]["others"].items()})
PY
done
