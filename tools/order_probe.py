#!/usr/bin/env python3
"""What would a spatially coherent internal point order be worth?  The reference shuffles a tile's points (random_sample takes prefixes), so every neighbour
gather of the pyramid and the network reads rows scattered over the tile.  Here the SAME tiles are fed in the order (pyramid band, Morton code) — every level's
point set is still a prefix — and the KNN pyramid + network are timed per kernel family in both orders (sequential, ssdr_prof)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al._lib import DevArray
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
hp._front_end(); _lib.sync()
B, N = hp.B, ConfigS3DIS.num_points
xyz = hp.xyz.to_host().reshape(B, N, 3).copy(); feat = hp.feat.to_host().reshape(B, N, -1).copy()

def timed(tag):
    for _ in range(2):
        hp._pyramid(); hp._infer(); _lib.sync()
    L.ssdr_prof_enable(1)
    for _ in range(5):
        hp._pyramid(); hp._infer()
    _lib.sync()
    rep = L.ssdr_prof_report().decode().strip().splitlines(); L.ssdr_prof_enable(0)
    rows = {}
    for ln in rep:
        name, calls, ms, work, work2 = ln.rsplit(" ", 4)
        rows[name] = float(ms) / 5
    print("%s: total %.3f ms per step" % (tag, sum(rows.values())))
    return rows

a = timed("reference order (shuffled)")
ratios = ConfigS3DIS.sub_sampling_ratio
Ns = [N]
for r in ratios: Ns.append(Ns[-1] // r)
band = np.zeros(N, np.int64)
for l in range(1, len(Ns)): band[:Ns[l]] = l            # deepest prefix a point belongs to
def morton(p):
    q = ((p - p.min(0)) / (np.ptp(p, 0).max() + 1e-9) * 1023).astype(np.int64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
for b in range(B):
    perm = np.lexsort((morton(xyz[b]), -band))          # deepest band first, Morton inside a band: every level's set stays a prefix
    xyz[b] = xyz[b][perm]; feat[b] = feat[b][perm]
hp.xyz = DevArray.from_host(xyz.reshape(B * N, 3).astype(np.float32)); hp.feat = DevArray.from_host(feat.reshape(B * N, -1).astype(np.float32))
s = timed("(band, Morton) order")
print("%-34s %10s %10s" % ("site", "shuffled", "sorted"))
for k in sorted(a, key=lambda k: -a[k]):
    print("%-34s %10.4f %10.4f" % (k, a[k], s.get(k, 0.0)))
