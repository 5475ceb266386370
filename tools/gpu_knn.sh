#!/bin/bash
# KNN GPU check: parity tests, bench stage times, per-kernel stats (sequential)
OUT=$GRAFT_REPO_ROOT/gpurun_out/knn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_knn.py tests/test_properties.py tests/test_pipeline.py -m gpu -x -q 2>&1 | tail -3
SSDR_KNN_DEBUG=1 python tools/knn_dbg.py 2>&1 | head -3
P=${1:-bf16x3}
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --stages --precision $P > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json")); print("$P", d["value"], d["ms_per_step"], d["stage_ms"], {k:v["ms_per_step"] for k,v in d["roofline"]["others"].items()})
PY
tail -3 $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --precision $P > /dev/null 2> $OUT/kts.err
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kts/kts_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "grid_" in n or "kd_" in n:
        short=n.replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
        per=float(r["TotalDurationNs"])/19.0/1e3
        tot+=per
        print("  %-46s calls/step %5.1f  avg %8.1f us  per-step %8.1f us"%(short, int(r["Calls"])/19.0, float(r["AverageNs"])/1e3, per))
print("  knn kernels per step: %.1f us" % tot)
PY
cp $OUT/kts/kts_kernel_stats.csv $OUT/seq_kernel_stats.csv
rm -rf $OUT/kts
