#!/bin/bash
# round 6: the split-row chain with one XCD's workgroups (SSDR_FPS_TEAM=1: records through that XCD's L2) against chip-wide
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6d.txt; : > $out
for nc in "20000 10000" "24000 10000" "9472 4736" "2368 1184"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --save /tmp/seq_$1.npy >> $out 2>&1
  for team in 0 1; do
    SSDR_FPS_TEAM=$team SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy >> $out 2>&1
  done
done
SSDR_FPS_TEAM=1 SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out
SSDR_FPS_TEAM=1 SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py 9472 4736 2>&1 | tail -9 >> $out
cat $out
