#!/bin/bash
# rocprofv3 --stats over many different batches of the AL round's inference half: kernels whose slowest call is far above their mean
OUT=$GRAFT_REPO_ROOT/gpurun_out/almany
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 tools/al_many_probe.py ${1:-100} ${2:-5000} > $OUT/run.txt 2> $OUT/kt.err
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_stats.csv")))
def nm(n): return n.replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
print("%-44s %6s %9s %9s %9s %9s" % ("kernel","calls","total ms","avg us","min us","max us"))
for r in sorted(rows, key=lambda r:-float(r["MaxNs"]))[:16]:
    print("%-44s %6d %9.2f %9.1f %9.1f %9.1f" % (nm(r["Name"])[:44], int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $OUT/kt; tail -4 $OUT/run.txt
