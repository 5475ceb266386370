#!/bin/bash
# round 6: rows over FOUR lanes (SSDR_FPS_COOP_SWEEP=5: 25 dependent instructions, twice the workgroups) with the first polling pass delayed, beside the default (two lanes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps_delay4.txt; : > $out
for nc in "9472 4736" "20000 10000" "24000 10000"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --save /tmp/seq_$1.npy 2>&1 | tr '\n' ' ' >> $out; echo >> $out
  for rep in 1 2; do
    echo -n "n=$1 form 4 default delay: " >> $out
    timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy 2>&1 | tr '\n' ' ' | sed 's/ssdr_fps_dev //; s/sequence identical to .*: /same=/' >> $out; echo >> $out
    for d in 12 16 20 24 28; do
      echo -n "n=$1 form 5 delay $d: " >> $out
      SSDR_FPS_COOP_SWEEP=5 SSDR_FPS_DELAY=$d timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy 2>&1 | tr '\n' ' ' | sed 's/ssdr_fps_dev //; s/sequence identical to .*: /same=/' >> $out; echo >> $out
  done; done
done
SSDR_FPS_COOP_SWEEP=5 SSDR_FPS_DELAY=20 SSDR_FPS_DBG=1 timeout 300 python3 tools/fps_large.py 20000 10000 2>&1 | tail -9 >> $out
cat $out
