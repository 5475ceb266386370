#!/bin/bash
# round 6: the split-row chain with two polling passes in flight (SSDR_FPS_STAGGER = s_sleep units between them)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6e.txt; : > $out
SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py 20000 10000 --save /tmp/seq.npy >> $out 2>&1
for st in 0 2 4 6 8 12 16; do
  echo "stagger $st" >> $out
  SSDR_FPS_STAGGER=$st SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py 20000 10000 --cmp /tmp/seq.npy >> $out 2>&1
done
for st in 0 6; do SSDR_FPS_STAGGER=$st SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out; done
for nc in "24000 10000" "9472 4736" "4736 2368" "2368 1184"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --save /tmp/seq_$1.npy >> $out 2>&1
  for st in 0 6; do SSDR_FPS_STAGGER=$st SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy >> $out 2>&1; done
done
cat $out
