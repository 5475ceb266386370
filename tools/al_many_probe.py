#!/usr/bin/env python3
"""many different batches through the AL round's inference half (for a rocprofv3 --stats run: kernels whose slowest call is far above their mean)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(int(sys.argv[2]) + i if len(sys.argv) > 2 else 5000 + i, density=5000.0) for i in range(16)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ar = pipeline.ALRound(W, rooms, nb, ConfigS3DIS, batch_size=10000, precision="bf16x3")
ar.infer_all()
for s in ar.streams + ar.bstreams: _lib.sync(s)
_lib.sync()
from ssdr_al import knn as _knn
for st in ar.bstreams:
    print("knn status", _knn.knn_status(st))
