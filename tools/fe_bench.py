#!/usr/bin/env python3
"""Front end alone (grid subsample + tile of the bench's 16 rooms), both implementations of the batched grid subsample:
   python tools/fe_bench.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
methods = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 0, 1, 0]
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
ref = None
for method in methods:
    hp.subsample_method = method
    hp._front_end(); _lib.sync()
    t0 = time.perf_counter()
    for _ in range(it):
        hp._front_end()
    _lib.sync()
    dt = (time.perf_counter() - t0) / it * 1e3
    m = hp.sub_m.to_host()[:16]
    out = hp.xyz.to_host()
    if ref is None:
        ref = out
    print("method %s: %.3f ms per front end (%d raw points -> %d voxels), tiles identical to the first run: %s"
          % ("sort" if method else "partition", dt, int(hp.room_off[-1]), int(m.sum()), bool(np.array_equal(out, ref))))
