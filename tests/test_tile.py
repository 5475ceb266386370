"""Tile generator (scope row N1): spatially_regular_gen of the sampling / test datasets
(s3dis_dataset.py:115-154, s3dis_dataset_test.py:105-143) against a NumPy restatement."""
import ctypes as C

import numpy as np

from conftest import assert_bits_equal


def _ref_tile(points, colors, pick, perm, num_points, possibility):
    d = points - pick[None]
    dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]          # float32, as the sort keys
    order = np.argsort(dist, kind="stable")[:num_points]                          # KDTree.query(pick, k=num_points)
    queried = order[perm]                                                         # DP.shuffle_idx with the caller's permutation
    xyz = points[queried] - pick[None]
    feat = np.concatenate([xyz, colors[queried] * np.float32(1 / 255.0)], 1)
    dists = np.sum(np.square((points[queried] - pick[None]).astype(np.float32)), axis=1)     # test.py:140
    delta = np.square(1 - dists / np.max(dists))                                            # :141
    possibility = possibility.copy()
    possibility[queried] += delta                                                           # :142
    return xyz, feat, queried, possibility


def test_tile_select_and_possibility_update(backend):
    from ssdr_al import _lib
    from ssdr_al._lib import DevArray
    rng = np.random.default_rng(12)
    n, N = (5000, 1024) if backend == "emu" else (90000, 40960)
    pts = (rng.random((n, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.float32)
    poss = rng.random(n) * 1e-3                                                   # init_possibility (test.py:84-91)
    pick = (pts[int(np.argmin(poss))] + rng.normal(0, 0.35, 3)).astype(np.float32)
    perm = rng.permutation(N).astype(np.int32)
    d_p, d_c, d_m = DevArray.from_host(pts), DevArray.from_host(col), DevArray.from_host(np.array([n, 0], np.int64))
    d_perm, d_dup = DevArray.from_host(perm), DevArray.from_host(rng.random(N).astype(np.float32))
    d_xyz, d_feat, d_idx = DevArray((N, 3), np.float32), DevArray((N, 6), np.float32), DevArray((N,), np.int32)
    d_poss, d_min, d_arg = DevArray.from_host(poss), DevArray((1,), np.float64), DevArray((1,), np.int32)
    _lib.check(_lib.lib().ssdr_tile_select_possibility_dev(d_p.ptr, d_c.ptr, 3, d_m.ptr, n, _lib.ptr(pick), N, d_perm.ptr, d_dup.ptr, 1.0 / 255.0,
                                                          d_xyz.ptr, d_feat.ptr, d_idx.ptr, d_poss.ptr, d_min.ptr, d_arg.ptr, None))
    _lib.sync()
    xyz, feat, queried, poss_ref = _ref_tile(pts, col, pick, perm, N, poss)
    assert_bits_equal(d_xyz.to_host(), xyz, "tile xyz")
    assert_bits_equal(d_feat.to_host(), feat, "tile features")
    assert np.array_equal(d_idx.to_host(), queried)
    assert np.array_equal(d_poss.to_host(), poss_ref)                             # float64 map, bit-exact
    assert d_min.to_host()[0] == poss_ref.min() and d_arg.to_host()[0] == int(np.argmin(poss_ref))
