#!/bin/bash
# round 6: per-wave records (form 6) against the split form with a line per record (form 4, shift 6)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6c.txt; : > $out
for nc in "20000 10000" "24000 10000" "9472 4736" "4736 2368" "2368 1184"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --save /tmp/seq_$1.npy >> $out 2>&1
  SSDR_FPS_SLOT_SHIFT=6 SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy >> $out 2>&1
  SSDR_FPS_COOP_SWEEP=6 timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy >> $out 2>&1
done
SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=6 timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out
cat $out
