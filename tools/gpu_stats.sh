#!/bin/bash
# sequential per-kernel stats of the bench step: tools/gpu_stats.sh "<substr> <substr> ..." [precision] [pytest args]
F="$1"; P=${2:-bf16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$3" ]; then timeout 900 python -m pytest $3 -m gpu -x -q 2>&1 | tail -3; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kts -o kts -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipeline --precision $P > $OUT/bench_seq.json 2> $OUT/kts.err
python - <<PY
import csv, json
d=json.load(open("$OUT/bench_seq.json")); print("sequential:", d["value"], "Mpts/s", d["ms_per_step"], "ms/step", d["stage_ms"])
rows=list(csv.DictReader(open("$OUT/kts/kts_kernel_stats.csv")))
flt="$F".split()
steps=0
for r in rows:
    if "fps_block" in r["Name"] or "fps_step" in r["Name"]: steps=max(steps,int(r["Calls"]))
steps=steps or 17
tot=0; alltot=0
for r in rows:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
    per=float(r["TotalDurationNs"])/steps/1e3
    alltot+=per
    if flt and not any(f in n for f in flt): continue
    tot+=per
    if per>=3: print("  %-46s calls/step %5.1f  avg %8.1f us  per-step %8.1f us"%(n[:46], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, per))
print("  selected kernels per step: %.1f us of %.1f us (all kernels)" % (tot, alltot))
PY
rm -rf $OUT/kts
