#!/bin/bash
# round 6: the split-row chain, the first polling pass delayed after the store: adapted per workgroup (default) against fixed delays (SSDR_FPS_DELAY, s_sleep units)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/fps_delay3.txt; : > $out
for nc in "2368 1184" "9472 4736" "20000 10000" "24000 10000" "65000 8000"; do
  set -- $nc
  SSDR_FPS_COOP_SWEEP=0 timeout 300 python3 tools/fps_large.py $1 $2 --save /tmp/seq_$1.npy 2>&1 | tr '\n' ' ' >> $out; echo >> $out
  for rep in 1 2; do for d in adaptive 0 16 20 24; do
    echo -n "n=$1 delay $d: " >> $out
    if [ $d = adaptive ]; then unset SSDR_FPS_DELAY; else export SSDR_FPS_DELAY=$d; fi
    timeout 300 python3 tools/fps_large.py $1 $2 --cmp /tmp/seq_$1.npy 2>&1 | tr '\n' ' ' | sed 's/ssdr_fps_dev //; s/sequence identical to .*: /same=/' >> $out; echo >> $out
  done; done
  unset SSDR_FPS_DELAY
  SSDR_FPS_DBG=1 timeout 300 python3 tools/fps_large.py $1 $2 2>&1 | grep -E "sweep|passes|polling" >> $out
done
cat $out
