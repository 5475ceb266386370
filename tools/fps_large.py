#!/usr/bin/env python3
"""FPS at the reference's real scale (ssdr_main_S3DIS2.py:134: 10 000 picks over ~2 x 10^4 candidate regions, 32-d features):
   python tools/fps_large.py [n] [count]   -> time of ssdr_fps_dev, and the index sequence against the NumPy oracle"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib
from ssdr_al._lib import DevArray
from oracle import select_np as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
_lib.check(_lib.lib().ssdr_init(0))
f = np.random.default_rng(1).normal(size=(n, 32))
d_f = DevArray.from_host(f); d_o = DevArray((count,), np.int32)
for rep in range(2):
    _lib.sync(); t0 = time.perf_counter()
    _lib.check(_lib.lib().ssdr_fps_dev(d_f.ptr, n, 32, 0, count, d_o.ptr, None))
    _lib.sync(); dt = time.perf_counter() - t0
    print("ssdr_fps_dev n=%d count=%d: %.1f ms (%.2f us per pick)" % (n, count, dt * 1e3, dt * 1e6 / count))
got = d_o.to_host()
t0 = time.perf_counter(); exp = np.asarray(S.farthest_features_sample(f, count, 0)); t1 = time.perf_counter() - t0
print("oracle (NumPy): %.1f s; sequences identical: %s" % (t1, bool(np.array_equal(got, exp))))
