#!/usr/bin/env python3
"""one very large raw cloud through the front end (Semantic3D scenes hold tens of millions of points): time per point beside the bench's rooms, and which formulation ran"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
for dens, nrooms in ((5000.0, 16), (60000.0, 2), (150000.0, 1)):
    rooms = [synthetic.make_room(5000 + i, density=dens) for i in range(nrooms)]
    hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
    for _ in range(2): hp._front_end(); _lib.sync()
    L.ssdr_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(3): hp._front_end()
    _lib.sync(); dt = (time.perf_counter() - t0) / 3
    rep = L.ssdr_prof_report().decode().strip().splitlines(); L.ssdr_prof_enable(0)
    rows = {ln.rsplit(" ", 4)[0]: float(ln.rsplit(" ", 4)[2]) / 3 for ln in rep}
    n = sum(len(r[0]) for r in rooms)
    print("%d room(s), %.1f M raw points: front end %.3f ms = %.1f ps per point; %s" % (nrooms, n / 1e6, dt * 1e3, dt * 1e12 / n, {k: round(v, 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:5]}))
    _lib.check(L.ssdr_grid_subsample_status(None, None))
