#!/bin/bash
# the -m gpu suite N times in a row on one box (flakiness check); the first failure's output is kept
N=${1:-5}
mkdir -p gpurun_out/soak
: > gpurun_out/soak/summary.txt
for i in $(seq 1 $N); do
  python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/soak/run_$i.txt 2>&1
  echo "run $i: rc=$? $(tail -1 gpurun_out/soak/run_$i.txt)" >> gpurun_out/soak/summary.txt
done
cat gpurun_out/soak/summary.txt
