#!/usr/bin/env python3
"""heavy-tailed superpoint sizes (real partitions have floors and walls of thousands of points; the synthetic stand-in's coarse-grid blobs hold ~90): what the
selection stage costs when some candidates are large — sequential stage times and the selection families, uniform blobs against blobs + planar slabs"""
import os, sys
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
    big = float(os.environ.get("SP_BIG", "1.5"))
    floor = xyz[:, 2] - lo[2] < 0.15
    kf = np.floor((xyz - lo) / big).astype(np.int64)
    key = np.where(floor, (1 << 40) + kf[:, 0] + 4096 * kf[:, 1], key)
    wall = (~floor) & (xyz[:, 0] - lo[0] < 0.15)
    key = np.where(wall, (2 << 40) + kf[:, 1] + 4096 * kf[:, 2], key)
    order = np.argsort(key, kind="stable"); ks = key[order]
    heads = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    return np.concatenate([heads, [len(ks)]]).astype(np.int32), order.astype(np.int32)

def run(tag):
    hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
    sizes = np.diff(hp.sp_off.to_host()) if hasattr(hp.sp_off, "to_host") else None
    for _ in range(2): hp.step()
    hp.step(timed_stages=True)
    L.ssdr_prof_enable(1)
    for _ in range(3): hp.step()
    _lib.sync()
    rep = L.ssdr_prof_report().decode().strip().splitlines(); L.ssdr_prof_enable(0)
    rows = {}
    for ln in rep:
        name, calls, ms, work, work2 = ln.rsplit(" ", 4)
        rows[name] = float(ms) / 3
    print("%s: %d superpoints, sizes mean %.0f max %d (>640: %d, >2000: %d); stages %s" % (tag, len(sizes), sizes.mean(), sizes.max(), (sizes > 640).sum(), (sizes > 2000).sum(),
          {k: round(float(v), 3) for k, v in hp.timing.items()}))
    print("   " + ", ".join("%s %.3f" % (k, v) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]) if k.startswith(("sel_", "fps", "cand"))))

run("coarse-grid blobs")
synthetic.superpoints_from_tile = slabs
run("blobs + floor / wall slabs")
