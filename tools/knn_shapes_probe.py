#!/usr/bin/env python3
"""the KNN entry points at the shapes the reference's OTHER call sites use (beside the pyramid): the projection of every raw point on the subsampled cloud
(data_prepare: knn_search(sub_xyz, xyz, 1), ~1 M queries on ~150 k points) and a whole subsampled cloud against itself with K = 16"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, knn, synthetic, subsampling
L = _lib.lib(); _lib.check(L.ssdr_init(0))
raw = synthetic.make_room(5003, density=5000.0)[0]
sub = subsampling.compute(raw, sampleDl=0.04)
if isinstance(sub, tuple): sub = sub[0]
sub = np.ascontiguousarray(sub, np.float32)
print("raw %d points, subsampled %d" % (len(raw), len(sub)))
for name, sup, qry, k in (("projection K=1", sub, raw, 1), ("cloud on itself K=16", sub, sub, 16), ("raw on itself K=16", raw, raw, 16)):
    knn.knn_batch(sup[None], qry[None], k); _lib.sync()
    t0 = time.perf_counter(); out = knn.knn_batch(sup[None], qry[None], k); dt = time.perf_counter() - t0
    print("%-24s support %8d queries %8d: %.2f ms (host to host), status %s" % (name, len(sup), len(qry), dt * 1e3, knn.knn_status()))
