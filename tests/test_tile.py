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


def test_tile_select_batch_ties_and_crowded_distance_bins(backend):
    """The batch flavour sorts only the rows that can be among the nearest num_points (distance histogram over the top 12 bits of the float
    pattern, candidates compacted, sorted by distance bits, runs of equal distances put back into index order).  Shapes that leave its
    common case: rows at EXACTLY equal distances (ties go by index), a shell of thousands of rows inside one histogram bin, a room smaller
    than the tile."""
    from ssdr_al import _lib
    from ssdr_al._lib import DevArray
    rng = np.random.default_rng(3)
    N = 1024 if backend == "emu" else 8192
    clouds, centers = [], []
    # (a) a lattice around its centre: many equal distances
    g = np.stack(np.meshgrid(*[np.arange(-12, 13)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * np.float32(0.25)
    clouds.append(g[rng.permutation(len(g))]); centers.append(np.zeros(3, np.float32))
    # (b) a thin shell: thousands of rows within 1 % of one radius, plus a sparse cloud inside
    n_shell = 3000 if backend == "emu" else 9000
    d = rng.normal(size=(n_shell, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    shell = (d * (2.0 + rng.random((n_shell, 1)) * 0.01)).astype(np.float32)
    inner = (rng.random((200 if backend == "emu" else 2000, 3)) * 2 - 1).astype(np.float32)
    clouds.append(np.concatenate([inner, shell])[rng.permutation(len(inner) + n_shell)]); centers.append(np.array([0.01, -0.02, 0.005], np.float32))
    # (c) fewer rows than the tile
    clouds.append((rng.random((N // 3, 3), dtype=np.float32) * 4).astype(np.float32)); centers.append(np.array([2, 2, 2], np.float32))
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int64)
    P = np.concatenate(clouds); col = rng.integers(0, 256, (len(P), 3)).astype(np.float32)
    R = len(clouds)
    perm = np.stack([rng.permutation(N) for _ in range(R)]).astype(np.int32); dup = rng.random((R, N)).astype(np.float32)
    d_p, d_c = DevArray.from_host(P), DevArray.from_host(col)
    d_m = DevArray.from_host(np.array([len(c) for c in clouds] + [0], np.int64))
    d_perm, d_dup = DevArray.from_host(perm), DevArray.from_host(dup)
    d_xyz, d_feat, d_idx = DevArray((R, N, 3), np.float32), DevArray((R, N, 6), np.float32), DevArray((R, N), np.int32)
    cen = np.ascontiguousarray(np.stack(centers), np.float32)
    lab = rng.integers(0, 13, len(P)).astype(np.int32)
    d_lab, d_olab = DevArray.from_host(lab), DevArray((R, N), np.int32)
    _lib.check(_lib.lib().ssdr_tile_select_batch_dev(d_p.ptr, d_c.ptr, 3, d_m.ptr, _lib.ptr(off), R, _lib.ptr(cen), N, d_perm.ptr, d_dup.ptr, 1.0 / 255.0,
                                                    d_xyz.ptr, d_feat.ptr, d_idx.ptr, d_lab.ptr, d_olab.ptr, None))
    _lib.sync()
    got = d_idx.to_host()
    # queried_pc_label = input_label[queried_idx] (s3dis_dataset.py:141): the labels travel with the rows, duplicates included
    assert np.array_equal(d_olab.to_host(), np.stack([lab[off[r]:off[r + 1]][got[r]] for r in range(R)]))
    for r in range(2):                      # full tiles: index for index
        pts = clouds[r]; dd = pts - cen[r][None]
        dist = (dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2]
        order = np.argsort(dist, kind="stable")[:N]
        assert np.array_equal(got[r], order[perm[r]]), "cloud %d" % r
    # the small room: its rows, shuffled, then duplicates of them (data_aug)
    m = len(clouds[2])
    assert set(got[2].tolist()) == set(range(m))
    # ... index for index: the entries of the permutation below m, in order, shuffle the m rows (nearest first); row r >= m repeats entry floor(u_r * m) of that list
    dd = clouds[2] - cen[2][None]
    order = np.argsort((dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1]) + dd[:, 2] * dd[:, 2], kind="stable")
    shuffled = order[perm[2][perm[2] < m]]
    pick = np.minimum((dup[2, m:] * np.float32(m)).astype(np.int64), m - 1)
    assert np.array_equal(got[2], np.concatenate([shuffled, shuffled[pick]]))
