#!/bin/bash
# A/B of the network kernels: sequential bench under the kernel trace, one run per environment setting given as arguments
# (e.g. "SSDR_LFA32=0" "SSDR_LFA32=1" "SSDR_LFA32_PF=1"); the first run also does the GPU parity tests of the network
OUT=$GRAFT_REPO_ROOT/gpurun_out/lfa_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_randla.py -m gpu -x -q -s 2>&1 | tail -12
i=0
for v in "$@"; do
  i=$((i+1))
  env $v python3 -c "print('== $v')"
  ( export $v; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts$i -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline $BENCH_EXTRA > $OUT/bench_seq_$i.json 2> $OUT/kts$i.err )
  python3 - <<PY
import csv, json
rows=list(csv.DictReader(open("$OUT/kts$i/kts_kernel_stats.csv")))
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
    j=json.loads(open("$OUT/bench_seq_$i.json").read().strip().splitlines()[-1]); print("  stage_ms", j["stage_ms"], "value", j["value"])
except Exception as e: print("bench line:", e)
PY
  rm -rf $OUT/kts$i
done
