#!/bin/bash
# the replicated global FPS of 2 / 4 / 8 ranks on one GPU (rows in registers, cooperative workgroups): per-pick time of the two hand-off forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for form in 0 1; do
  echo "== SSDR_FPS_COOP_COUNTER=$form"
  for nc in "2368 1184" "4736 2368" "9472 4736" "20000 10000"; do SSDR_FPS_COOP_COUNTER=$form python3 tools/fps_large.py $nc 2>&1 | grep -E "ssdr_fps_dev|identical" | tail -2; done
done
