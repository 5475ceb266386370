"""Scratch timing of the network stage."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import ctypes as C
from ssdr_al import _lib, randlanet
from oracle import randla_np as R
rng = np.random.default_rng(0)
B, N = 16, 40960
xyz = (rng.random((B, N, 3), dtype=np.float32) * np.array([10, 8, 3], np.float32)).astype(np.float32)
feat = np.concatenate([xyz - xyz.mean(1, keepdims=True), rng.random((B, N, 3), dtype=np.float32)], -1)
net = randlanet.Network().load(R.init_weights(0))
cfg = net.config; L, K = 5, 16
sizes = [N]
for r in cfg.sub_sampling_ratio: sizes.append(sizes[-1] // r)
d_xyz = _lib.DevArray.from_host(xyz); d_f = _lib.DevArray.from_host(feat)
neigh = [_lib.DevArray((B, sizes[i], K), np.int32) for i in range(L)]
interp = [_lib.DevArray((B, sizes[i], 1), np.int32) for i in range(L)]
arr = C.c_void_p * L
r = np.asarray(cfg.sub_sampling_ratio, np.int32)
probs = _lib.DevArray((B * N, 13), np.float32); f32 = _lib.DevArray((B * N, 32), np.float32)
for it in range(4):
    _lib.sync(); t0 = time.time()
    _lib.check(_lib.lib().ssdr_knn_pyramid_dev(d_xyz.ptr, B, N, L, _lib.ptr(r), K, arr(*[a.ptr for a in neigh]), None, arr(*[a.ptr for a in interp]), None))
    _lib.sync(); t1 = time.time()
    net.infer_dev(B, N, d_f.ptr, d_xyz.ptr, [a.ptr for a in neigh], [a.ptr for a in interp], probs.ptr, f32.ptr)
    _lib.sync(); t2 = time.time()
    print("pyramid %.2f ms  infer %.2f ms  -> infer %.1f Mpts/s, both %.1f Mpts/s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, B * N / (t2 - t1) / 1e6, B * N / (t2 - t0) / 1e6))
