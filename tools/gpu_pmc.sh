#!/bin/bash
# PMC passes (SQ counters) over a sequential bench step: tools/gpu_pmc.sh <precision> "<counters pass 1>" ["<pass 2>" ...]
P=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$P
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ -f $OUT/counters.txt ] || rocprofv3 -L > $OUT/counters.txt 2>&1
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --precision $P > /dev/null 2> $OUT/p$i.err
  cp $OUT/p$i/p_counter_collection.csv $OUT/pass$i.csv; rm -rf $OUT/p$i
done
ls -la $OUT
