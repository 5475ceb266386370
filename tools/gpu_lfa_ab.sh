#!/bin/bash
# A/B of the network kernels: sequential bench under the kernel trace with the 16 x 16-tile LFA kernels (SSDR_LFA32=0) and the 32 x 32-tile ones
OUT=$GRAFT_REPO_ROOT/gpurun_out/lfa_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_randla.py -m gpu -x -q -s 2>&1 | tail -12
for v in 0 1; do
  export SSDR_LFA32=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts$v -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline > $OUT/bench_seq_$v.json 2> $OUT/kts$v.err
  echo "== SSDR_LFA32=$v"; python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$OUT/kts$v/kts_kernel_stats.csv")))
tot=0
for r in rows:
    n=r["Name"]
    if "lfa" in n or "dense" in n or "tail" in n or "gather_max" in n:
        short=n.split("(")[0].replace("void ssdr::","").replace("ssdr::","")
        per=float(r["TotalDurationNs"])/19.0/1e3
        tot+=per
        print("  %-46s calls/step %5.1f  avg %8.1f us  per-step %8.1f us"%(short, int(r["Calls"])/19.0, float(r["AverageNs"])/1e3, per))
print("  network kernels per step: %.1f us" % tot)
try:
    j=json.loads(open("$OUT/bench_seq_$v.json").read().strip().splitlines()[-1]); print("  stage_ms", j["stage_ms"], "value", j["value"])
except Exception as e: print("bench line:", e)
PY
  rm -rf $OUT/kts$v
done
