#!/usr/bin/env python3
"""the AL round with a heavy-tailed partition (floor / wall slabs beside the coarse-grid blobs, as tools/sp_probe.py) against the stand-in's"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
orig = synthetic.superpoints_from_tile
def slabs(xyz, cell=0.3):
    lo = xyz.min(0)
    k = np.floor((xyz - lo) / cell).astype(np.int64)
    key = k[:, 0] + 4096 * (k[:, 1] + 4096 * k[:, 2])
    kf = np.floor((xyz - lo) / 1.5).astype(np.int64)
    floor = xyz[:, 2] - lo[2] < 0.15
    key = np.where(floor, (1 << 40) + kf[:, 0] + 4096 * kf[:, 1], key)
    wall = (~floor) & (xyz[:, 0] - lo[0] < 0.15)
    key = np.where(wall, (2 << 40) + kf[:, 1] + 4096 * kf[:, 2], key)
    order = np.argsort(key, kind="stable"); ks = key[order]
    heads = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    return np.concatenate([heads, [len(ks)]]).astype(np.int32), order.astype(np.int32)
for tag, fn in (("coarse-grid blobs", orig), ("blobs + floor / wall slabs", slabs)):
    synthetic.superpoints_from_tile = fn
    ar = pipeline.ALRound(W, rooms, 17, ConfigS3DIS, batch_size=10000, precision="bf16x3")
    ar.run(); _lib.sync()
    L.ssdr_prof_enable(1)
    t0 = time.perf_counter(); ar.run(); _lib.sync(); dt = time.perf_counter() - t0
    rep = L.ssdr_prof_report().decode().strip().splitlines(); L.ssdr_prof_enable(0)
    rows = {ln.rsplit(" ", 4)[0]: float(ln.rsplit(" ", 4)[2]) for ln in rep}
    print("%s: %d regions; round %.1f ms; %s" % (tag, ar.sel.S, dt * 1e3, ", ".join("%s %.2f" % (k, v) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]) if k.startswith(("sel_", "fps", "cand")))[:330]))
