#!/bin/bash
# round 6: where a pick's time goes in fps_coop_sweep (SSDR_FPS_DBG: s_memtime between the phases, wave 0 of every workgroup)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6_dbg.txt; : > $out
for nc in "20000 10000" "9472 4736" "2368 1184"; do
  set -- $nc
  for form in 1 2 3; do
    SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=$form timeout 300 python3 tools/fps_large.py $1 $2 >> $out 2>&1
  done
done
cat $out
