#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kc -o k -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipeline --selector kcenter > gpurun_out/kc.json 2>/dev/null
python - <<PY
import csv, json
d=json.loads([l for l in open("gpurun_out/kc.json") if l.startswith("{")][0]); print(d["value"], d["ms_per_step"], d["stage_ms"])
rows=list(csv.DictReader(open("gpurun_out/kc/k_kernel_stats.csv")))
steps=max(int(r["Calls"]) for r in rows if "kc_init" in r["Name"])
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:14]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
    print("  %-44s calls/step %5.1f avg %9.1f us per-step %9.1f us" % (n[:44], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/steps/1e3))
PY
rm -rf gpurun_out/kc
