#!/bin/bash
# round 6: the split-row chain's record slots on lines of their own (SSDR_FPS_SLOT_SHIFT: 4 = packed 16-byte slots, 7 = a 128-byte line each, 8, 12)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps6b.txt; : > $out
SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py 20000 10000 --save /tmp/seq.npy >> $out 2>&1
for form in 4 5; do for sh in 4 6 7 8 10 12; do
  echo "form $form slot shift $sh" >> $out
  SSDR_FPS_SLOT_SHIFT=$sh SSDR_FPS_COOP_SWEEP=$form timeout 300 python3 tools/fps_large.py 20000 10000 --cmp /tmp/seq.npy >> $out 2>&1
done; done
for sh in 7 12; do SSDR_FPS_SLOT_SHIFT=$sh SSDR_FPS_DBG=1 SSDR_FPS_COOP_SWEEP=4 timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out; done
cat $out
