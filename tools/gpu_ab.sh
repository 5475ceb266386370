#!/bin/bash
# Same-box A/B of builds: the boxes of the pool differ by +-3 %, so two numbers from two gpurun calls compare boxes, not code.  Build the variants here
# (hipcc cross-compiles), keep them under the repo (e.g. abtmp/lib_<name>.so: an untracked directory travels with the snapshot), then in ONE call:
#   gpurun -- 'bash tools/gpu_ab.sh abtmp/lib_a.so abtmp/lib_b.so'
# alternates the default bench over them three times (ssdr_al/_lib.py loads $SSDR_AL_LIBRARY instead of the in-tree library).
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    SSDR_AL_LIBRARY=$PWD/$lib timeout 120 python3 bench.py --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done
