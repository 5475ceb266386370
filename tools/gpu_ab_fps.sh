#!/bin/bash
# same-box A/B of two builds, the FPS chain and the selection stage shown: tools/gpu_ab_fps.sh abtmp/lib_a.so abtmp/lib_b.so
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in "$@"; do
  echo -n "bench $lib: "
  SSDR_AL_LIBRARY=$PWD/$lib timeout 120 python3 bench.py --no-cpu-baseline --steps 100 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['others']; print(d['value'], d['ms_per_step'], d['stage_ms']['select'], o['fps_chain']['ms_per_step'])"
done; done
