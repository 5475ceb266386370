#!/usr/bin/env python3
"""the selection's graph variants (gcn_number hops, gcn_top nearest per row) on the bench's rooms and in the AL round: no cliff beside the default (1 hop, dense adjacency)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
for gn, gt, sel in ((1, 0, "fps"), (3, 0, "fps"), (1, 100, "fps"), (3, 100, "fps"), (2, 10, "kcenter")):
    hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3", gcn_number=gn, gcn_top=gt, selector=sel).load_rooms(rooms)
    for _ in range(2): hp.step()
    hp.step(timed_stages=True)
    print("step gcn_number=%d gcn_top=%d %s: select %.3f ms" % (gn, gt, sel, hp.timing["select"]))
for gn, gt, sel in ((1, 0, "fps"), (3, 100, "fps"), (2, 10, "kcenter")):
    ar = pipeline.ALRound(W, rooms, 17, ConfigS3DIS, batch_size=10000, precision="bf16x3", selector=sel, gcn_number=gn, gcn_top=gt)
    ar.run(); _lib.sync()
    t0 = time.perf_counter(); ar.run(); _lib.sync(); dt = time.perf_counter() - t0
    print("AL round gcn_number=%d gcn_top=%d %s: %.2f ms" % (gn, gt, sel, dt * 1e3))
