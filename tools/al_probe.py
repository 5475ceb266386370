#!/usr/bin/env python3
"""where the inference half of an AL round spends its time: host enqueue alone, the three stages alone, overlapped"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 17
ar = pipeline.ALRound(W, rooms, nb, ConfigS3DIS, batch_size=10000, precision="bf16x3")
def sync():
    for s in ar.streams + ar.bstreams: _lib.sync(s)
    _lib.sync()
ar.infer_all(); sync()
for rep in range(2):
    t0 = time.perf_counter(); ar.infer_all(); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print("infer_all: host enqueue %.2f ms, total %.2f ms (%.2f per batch)" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / nb))
# stages alone, one batch at a time
for name in ("_front_end", "_pyramid", "_infer"):
    if len(sys.argv) > 2: break
    t0 = time.perf_counter()
    for b in range(nb):
        getattr(ar._bind(b), name)()
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print("%-10s x %d: host %.2f ms, total %.2f ms (%.3f per batch)" % (name, nb, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / nb))
