#!/bin/bash
# round 6: the hand-off forms of the cooperative FPS chain, same box: tools/gpu_fps6.sh [forms...]  (writes gpurun_out/fps6.txt)
# forms (SSDR_FPS_COOP_SWEEP): 0 rounds 3-5 (counter / polled granules), 1 swept records chip-wide, 2 / 3 one XCD's workgroups (plain / write-through stores),
# 4 / 5 rows split over 2 / 4 lanes with 16-byte records and the winner's row from the table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6.txt; : > $out
forms="${@:-1 4 5}"
for nc in "20000 10000" "24000 10000" "9472 4736" "4736 2368" "2368 1184"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --oracle --save /tmp/seq_$1.npy >> $out 2>&1
  for form in $forms; do
    SSDR_FPS_COOP_SWEEP=$form timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy >> $out 2>&1
  done
done
for form in 4 5; do SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=$form timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out; done
cat $out
