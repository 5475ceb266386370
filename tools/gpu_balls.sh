#!/bin/bash
# hand-over trees cut to the balls of the handed-over rows: fall-back rows per ball scale, then the usual KNN check
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for S in 1 2.25 4 9 16; do echo "scale $S"; SSDR_KNN_BALL_SCALE=$S SSDR_KNN_DEBUG=1 python tools/knn_dbg.py 2>&1 | head -1; done
bash tools/gpu_knn.sh
