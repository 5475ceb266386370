#!/usr/bin/env python3
"""AL round's inference half with more tiles per batch (the 16 raw rooms repeated): python tools/al_batch_probe.py tiles_per_batch n_batches"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
base = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
tpb, nb = int(sys.argv[1]), int(sys.argv[2])
rooms = [base[i % 16] for i in range(tpb)]
ar = pipeline.ALRound(W, rooms, nb, ConfigS3DIS, batch_size=10000, precision="bf16x3")
def sync():
    for s in ar.streams + ar.bstreams: _lib.sync(s)
    _lib.sync()
ar.infer_all(); sync()
for rep in range(2):
    t0 = time.perf_counter(); ar.infer_all(); sync(); t2 = time.perf_counter()
    print("tiles per batch %d x %d batches = %d tiles: infer_all %.2f ms (%.3f ms per 16 tiles)" % (tpb, nb, tpb * nb, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / (tpb * nb) * 16))
t0 = time.perf_counter(); out = ar.run(); t1 = time.perf_counter()
print("whole round %.2f ms, %d picks" % ((t1 - t0) * 1e3, len(out[0])))
