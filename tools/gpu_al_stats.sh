#!/bin/bash
# rocprofv3 --kernel-trace --stats of ONE process that runs the AL round twice (tools/al_sel_probe.py): the per-kernel summary for profiles/, and the selection half dispatch by dispatch
OUT=$GRAFT_REPO_ROOT/gpurun_out/alstats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 tools/al_sel_probe.py fps > $OUT/run.txt 2> $OUT/kt.err
cp $OUT/kt/kt_kernel_stats.csv $OUT/r06_al_round_kernel_stats.csv
rm -rf $OUT/kt
bash tools/gpu_al_sel.sh fps > /dev/null 2>&1
cp gpurun_out/alsel/alsel_fps.txt $OUT/r06_al_round_selection_dispatches.txt
head -12 $OUT/r06_al_round_kernel_stats.csv | cut -c1-160
