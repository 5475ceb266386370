#!/usr/bin/env python3
"""tiles per call other than the bench's 16: sequential stage times per tile (1, 3, 17, 32, 48 rooms)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
allrooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(48)]
for B in (1, 3, 16, 17, 32, 48):
    hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(allrooms[:B])
    for _ in range(2): hp.step()
    hp.step(timed_stages=True)
    tot = sum(hp.timing.values())
    print("B=%2d: %s  total %.3f ms = %.3f ms per tile" % (B, {k: round(float(v), 3) for k, v in hp.timing.items()}, tot, tot / B))
