#!/usr/bin/env python3
"""host-side cost of reading an AL round's selection back (pipeline.HotPath._select_collect): cProfile of the collect alone, GPU idle before it"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
ar = pipeline.ALRound(W, rooms, 17, ConfigS3DIS, batch_size=10000, precision="bf16x3")
ar.run(); _lib.sync()
for rep in range(3):
    ar.infer_all(); ar.sel._score_async(None); ar.sel._select_issue(None)
    for s in ar.streams + ar.bstreams: _lib.sync(s)
    _lib.sync()
    t0 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    out = ar.sel._select_collect()
    if rep == 2:
        pr.disable()
    t1 = time.perf_counter()
    from ssdr_al import knn as _knn
    for st in [ar.s_knn] + ar.bstreams:
        _knn.knn_status(st); _lib.check(_lib.lib().ssdr_grid_subsample_status(st, None))
    _lib.check(_lib.lib().ssdr_grid_subsample_status(ar.s_front, None))
    t2 = time.perf_counter()
    print("collect with the GPU idle: %.3f ms; the status reads of the five streams: %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
