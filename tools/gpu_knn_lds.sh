#!/bin/bash
# first pass of the grid KNN in its two forms on the bench batch: records from global memory (default) against the workgroup's LDS copy of the
# rows of cells its queries touch (SSDR_KNN_LDS=1, the north_star's formulation); parity tests under both, kernel times from the trace
OUT=$GRAFT_REPO_ROOT/gpurun_out/knn_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export SSDR_KNN_LDS=$v
  echo "== SSDR_KNN_LDS=$v"
  timeout 200 python -m pytest tests/test_knn.py -m gpu -x -q 2>&1 | tail -1
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts$v -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline > $OUT/bench_$v.json 2> $OUT/kts$v.err
  python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$OUT/kts$v/kts_kernel_stats.csv")))
for r in rows:
    n=r["Name"]
    if "grid_search" in n or "grid_retry" in n:
        short=n.replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
        print("  %-46s calls %4s  avg %8.1f us"%(short, r["Calls"], float(r["AverageNs"])/1e3))
j=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1]); print("  knn_pyramid stage ms:", j["stage_ms"]["knn_pyramid"])
PY
  rm -rf $OUT/kts$v
done
