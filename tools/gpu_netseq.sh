#!/bin/bash
# sequential bench under the kernel trace: the dispatches of ONE step in launch order, each with its median duration over the steps
P=${1:-bf16x3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/netseq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pipeline --precision $P > $OUT/bench.json 2> $OUT/kt.err
python - <<PY > $OUT/netseq_$P.txt
import csv, statistics
rows=list(csv.DictReader(open("$OUT/kt/kt_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n): return n.replace("(anonymous namespace)::","").replace("void ","").replace("ssdr::","").split("(")[0]
# steps are delimited by the first kernel of the front end (fe_minmax_partial)
starts=[i for i,r in enumerate(rows) if "fe_minmax_partial" in r["Kernel_Name"]]
steps=[rows[a:b] for a,b in zip(starts[:-1], starts[1:])]
steps=[s for s in steps if len(s)==len(steps[-1])]
print("steps with identical launch counts:", len(steps), "launches per step:", len(steps[-1]))
tot=0
for j in range(len(steps[-1])):
    ds=[int(s[j]["End_Timestamp"])-int(s[j]["Start_Timestamp"]) for s in steps]
    r=steps[-1][j]
    med=statistics.median(ds)/1e3
    tot+=med
    print("%4d %-44s grid %8s wg %4s lds %6s  %8.1f us" % (j, short(r["Kernel_Name"])[:44], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v","")), med))
print("sum %.1f us" % tot)
PY
tail -400 $OUT/netseq_$P.txt
rm -rf $OUT/kt
