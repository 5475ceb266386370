#!/usr/bin/env python3
"""the hot path in the Semantic3D flavour (helper_tool.py:77-117: 8 classes, 0.06 m grid, 65536-point tiles, float32 chamfer values) on synthetic rooms: stage times of a
sequential step and the pipelined step — a second configuration's sanity check beside the bench's S3DIS one"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigSemantic3D
L = _lib.lib(); _lib.check(L.ssdr_init(0))
W = synthetic.init_weights(0, num_classes=ConfigSemantic3D.num_classes)
rooms = [synthetic.make_room(6100 + i, density=9000.0) for i in range(16)]
print("raw points per room:", [len(r[0]) for r in rooms][:4], "...")
mk = lambda: pipeline.HotPath(W, ConfigSemantic3D, precision="bf16x3", chamfer_mode="f32_cuda").load_rooms(rooms)
hp = mk()
for _ in range(2): hp.step()
st = hp.step(timed_stages=True)
print("sequential stages (ms):", {k: round(float(v), 3) for k, v in hp.timing.items()})
pipe = pipeline.Pipelined(mk, 5)
pipe.run(6)
t0 = time.perf_counter(); pipe.run(60); dt = time.perf_counter() - t0
pipe.finish()
N = ConfigSemantic3D.num_points
print("pipelined: %.3f ms per step of 16 tiles x %d points = %.1f Mpoints/s" % (dt / 60 * 1e3, N, 16 * N * 60 / dt / 1e6))
L.ssdr_prof_enable(1)
for _ in range(3): hp.step()
_lib.sync()
rep = L.ssdr_prof_report().decode().strip().splitlines(); L.ssdr_prof_enable(0)
rows = []
for ln in rep:
    name, calls, ms, work, work2 = ln.rsplit(" ", 4)
    rows.append((float(ms) / 3, name))
for ms, name in sorted(rows, reverse=True)[:16]: print("  %-34s %8.3f ms" % (name, ms))
print("sum %.3f ms" % sum(r[0] for r in rows))
