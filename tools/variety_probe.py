#!/usr/bin/env python3
"""input variety the bench's 16 look-alike rooms do not have: rooms of very different sizes in one batch, a partition of many tiny regions — sequential stage
times and the largest kernel families of each case beside the bench's case"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0)
orig = synthetic.superpoints_from_tile

def run(tag, rooms):
    hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
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
    print("%s: raw points %s; %d regions; stages %s" % (tag, [len(r[0]) for r in rooms][:6], hp.S, {k: round(float(v), 3) for k, v in hp.timing.items()}))
    print("   " + ", ".join("%s %.3f" % (k, v) for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:9]))

base = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
run("bench rooms", base)
mixed = [synthetic.make_room(5000 + i, density=d) for i, d in enumerate([600, 900, 1500, 2500, 4000, 5000, 7000, 9000, 12000, 16000, 20000, 800, 3000, 6000, 10000, 14000])]
run("mixed densities", mixed)
synthetic.superpoints_from_tile = lambda xyz, cell=0.3: orig(xyz, 0.07)
run("tiny regions (0.07 m cells)", base)
