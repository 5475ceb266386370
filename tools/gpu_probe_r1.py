"""Scratch timing of the stages that exist so far (first GPU contact)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
from ssdr_al import _lib, knn, subsampling
rng = np.random.default_rng(0)
B, N = 16, 40960
xyz = (rng.random((B, N, 3), dtype=np.float32) * np.array([10, 8, 3], np.float32)).astype(np.float32)
xyz[:, : N // 2, 2] = 0
for it in range(3):
    t = time.time(); out = knn.knn_pyramid(xyz, [4, 4, 4, 4, 2], 16); w = time.time() - t
    print("pyramid B=16: wall %.1f ms, gpu %.3f ms -> %.1f Mpts/s (gpu)" % (w * 1e3, _lib.last_gpu_ms(), B * N / _lib.last_gpu_ms() / 1e3))
for it in range(3):
    t = time.time(); out = knn.knn_batch_i32(xyz, xyz, 16); w = time.time() - t
    print("knn16 B=16: wall %.1f ms, gpu %.3f ms" % (w * 1e3, _lib.last_gpu_ms()))
n = 1000000
pts = (rng.random((n, 3), dtype=np.float32) * np.array([10, 8, 3], np.float32)).astype(np.float32)
pts[: n // 2, 2] = rng.normal(0, 0.002, n // 2)
col = rng.integers(0, 256, (n, 3)).astype(np.float32); lab = rng.integers(0, 13, n).astype(np.int32)
for order in ("key", "reference"):
    for it in range(3):
        t = time.time(); o = subsampling.compute(pts, features=col, classes=lab, sampleDl=0.04, order=order); w = time.time() - t
        print("subsample 1M %s: M=%d wall %.1f ms gpu %.3f ms -> %.1f Mpts/s (gpu)" % (order, len(o[0]), w * 1e3, _lib.last_gpu_ms(), n / _lib.last_gpu_ms() / 1e3))
