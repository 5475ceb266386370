"""Grid-subsampling parity: HIP path through the C ABI / the reference-named module surface against the
golden vectors of the reference build and the oracle.  Bit-exact rows, reference row order included."""
import numpy as np
import pytest

from conftest import assert_bits_equal


def test_subsample_matches_reference_golden(backend, golden):
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling   # helper_tool.py:14
    g = golden("subsample_golden.npz")
    assert_bits_equal(cpp_subsampling.compute(g["hand/pts"], sampleDl=1.0), g["hand/out"], "hand order")
    for nm in ("tieA", "tieB"):
        out = cpp_subsampling.compute(g[nm + "/pts"], classes=g[nm + "/cls"], sampleDl=1.0)
        assert_bits_equal(out[1], g[nm + "/out_cls"], nm)
    p, f, c = cpp_subsampling.compute(g["room/pts"], features=g["room/col"], classes=g["room/lab"], sampleDl=0.04)
    assert p.dtype == np.float32 and f.dtype == np.float32 and c.dtype == np.int32 and c.ndim == 2
    assert_bits_equal(p, g["room/out_pts"]); assert_bits_equal(f, g["room/out_col"]); assert_bits_equal(c, g["room/out_lab"])
    assert_bits_equal(cpp_subsampling.compute(g["one/pts"], sampleDl=0.1), g["one/out"])
    assert_bits_equal(cpp_subsampling.compute(g["neg/pts"], sampleDl=0.3), g["neg/out"])
    p, c = cpp_subsampling.compute(g["manylab/pts"], classes=g["manylab/cls"], sampleDl=0.5)
    assert_bits_equal(p, g["manylab/out_pts"]); assert_bits_equal(c, g["manylab/out_cls"])


def test_subsample_interface_errors(backend):
    """Argument handling of wrapper.cpp:70-190."""
    import cpp_wrappers.cpp_subsampling.grid_subsampling as cpp_subsampling
    p = np.zeros((10, 3), np.float32)
    with pytest.raises(TypeError):
        cpp_subsampling.compute(p, p)                                   # features are keyword-only ("O|$OOfsi")
    with pytest.raises(RuntimeError, match="Error parsing method"):
        cpp_subsampling.compute(p, method="median")
    with pytest.raises(RuntimeError, match=r"points.shape is not \(N, 3\)"):
        cpp_subsampling.compute(np.zeros((10, 2), np.float32))
    with pytest.raises(RuntimeError, match=r"features.shape is not \(N, d\)"):
        cpp_subsampling.compute(p, features=np.zeros(10, np.float32))
    with pytest.raises(RuntimeError, match=r"features.shape is not \(N, d\)"):
        cpp_subsampling.compute(p, features=np.zeros((9, 2), np.float32))
    with pytest.raises(RuntimeError, match=r"classes.shape is not \(N,\) or \(N, d\)"):
        cpp_subsampling.compute(p, classes=np.zeros(9, np.int32))
    with pytest.raises(RuntimeError, match="^Error$"):
        cpp_subsampling.compute(np.zeros((0, 3), np.float32))
    out = cpp_subsampling.compute(p + 0.5, sampleDl=1.0, method="voxelcenters")   # accepted, still barycentres
    assert out.shape == (1, 3)


def test_data_processing_facade(backend, orc):
    """DataProcessing.grid_sub_sampling / knn_search keep their reference signatures (helper_tool.py:173-235)."""
    from ssdr_al.helper_tool import DataProcessing as DP
    rng = np.random.default_rng(2)
    n = 4000
    xyz = rng.random((n, 3), dtype=np.float32) * 2
    col = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    lab = rng.integers(0, 13, n).astype(np.uint8)
    sp, sc, sl = DP.grid_sub_sampling(xyz, col, lab, 0.04)                         # data_prepare_s3dis.py:58
    ep, ec, el = orc.grid_subsampling(xyz, col.astype(np.float32), lab.astype(np.int32), 0.04)
    assert_bits_equal(sp, ep); assert_bits_equal(sc, ec); assert_bits_equal(sl, el)
    assert DP.grid_sub_sampling(xyz, grid_size=0.1).shape[1] == 3
    idx = DP.knn_search(sp[None, :500], sp[None, :800], 16)
    assert idx.dtype == np.int32 and idx.shape == (1, 800, 16)
    assert_bits_equal(idx, orc.knn_batch(sp[None, :500], sp[None, :800], 16).astype(np.int32))


def test_subsample_fresh_inputs_against_oracle(backend, orc):
    from ssdr_al import subsampling
    rng = np.random.default_rng(13)
    n = 25000 if backend == "emu" else 1000000
    pts = (rng.random((n, 3), dtype=np.float32) * np.array([10, 8, 3], np.float32) - np.array([4, 3, 1], np.float32)).astype(np.float32)
    pts[: n // 2, 2] = np.float32(-1) + rng.normal(0, 0.002, n // 2).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.float32)
    lab = rng.integers(0, 13, n).astype(np.int32)
    dl = 0.06 if backend == "emu" else 0.04
    for order in ("reference", "key"):
        got = subsampling.compute(pts, features=col, classes=lab, sampleDl=dl, order=order)
        exp = orc.grid_subsampling(pts, col, lab, dl, order=order)
        for x, y in zip(got, exp):
            assert_bits_equal(x, y, order)


def test_subsample_batch_matches_per_cloud_oracle(backend, orc):
    """ssdr_grid_subsample_batch_dev: ragged clouds (one of a single point, one exactly a sort tile) in one call."""
    from ssdr_al import _lib
    from ssdr_al._lib import DevArray
    rng = np.random.default_rng(31)
    sizes = [1, 2048, 700, 5000, 33] if backend == "emu" else [1, 2048, 300000, 812345, 33, 40960]
    clouds = []
    for i, n in enumerate(sizes):
        p = (rng.random((n, 3), dtype=np.float32) * np.array([6, 5, 3], np.float32) + np.float32(i)).astype(np.float32)
        clouds.append((p, rng.integers(0, 256, (n, 3)).astype(np.float32), rng.integers(0, 13, (n, 1)).astype(np.int32)))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    P = np.concatenate([c[0] for c in clouds]); Fe = np.concatenate([c[1] for c in clouds]); Lb = np.concatenate([c[2] for c in clouds])
    d_p, d_f, d_l = DevArray.from_host(P), DevArray.from_host(Fe), DevArray.from_host(Lb)
    o_p, o_f, o_l = DevArray(P.shape, np.float32), DevArray(Fe.shape, np.float32), DevArray(Lb.shape, np.int32)
    d_m = DevArray((len(sizes),), np.int64)
    _lib.check(_lib.lib().ssdr_grid_subsample_batch_dev(d_p.ptr, d_f.ptr, 3, d_l.ptr, 1, _lib.ptr(off), len(sizes), 0.04, o_p.ptr, o_f.ptr, o_l.ptr, d_m.ptr, None))
    _lib.sync()
    m = d_m.to_host(); gp, gf, gl = o_p.to_host(), o_f.to_host(), o_l.to_host()
    for r, (p, f, l) in enumerate(clouds):
        ep, ef, el = orc.grid_subsampling(p, f, l, 0.04, order="key")
        assert m[r] == len(ep)
        s = int(off[r])
        assert_bits_equal(gp[s:s + m[r]], ep, "cloud %d points" % r)
        assert_bits_equal(gf[s:s + m[r]], ef, "cloud %d features" % r)
        assert_bits_equal(gl[s:s + m[r]], el, "cloud %d labels" % r)


@pytest.mark.parametrize("fdim,ldim,nlab", [(6, 2, 13), (3, 2, 40), (1, 1, 300), (4, 1, 13)])
def test_subsample_row_layouts(backend, orc, fdim, ldim, nlab):
    """Other feature / label widths than the hot path's (3, 1); labels outside [0,13) leave the fast majority-vote
    path (a 40/300-label voxel walks the unordered_map emulation, rehash included)."""
    from ssdr_al import subsampling
    rng = np.random.default_rng(100 * fdim + ldim)
    n = 6000 if backend == "emu" else 200000
    pts = (rng.random((n, 3), dtype=np.float32) * np.array([3, 2, 1], np.float32)).astype(np.float32)
    feat = rng.normal(0, 50, (n, fdim)).astype(np.float32)
    lab = (rng.integers(0, nlab, (n, ldim)) - (5 if nlab > 13 else 0)).astype(np.int32)
    dl = (0.2 if backend == "emu" else 0.065) if nlab > 13 else 0.08      # <= 29 distinct labels per voxel (LAB_CAP)
    for order in ("reference", "key"):
        got = subsampling.compute(pts, features=feat, classes=lab, sampleDl=dl, order=order)
        exp = orc.grid_subsampling(pts, feat, lab, dl, order=order)
        for x, y in zip(got, exp):
            assert_bits_equal(x, y, "%s fdim %d ldim %d" % (order, fdim, ldim))


def test_device_flavour_reports_label_table_overflow(backend):
    """A voxel with more distinct labels than the per-voxel table holds: the host flavour returns SSDR_ERR_UNSUPPORTED; the device flavour only
    enqueues, ssdr_grid_subsample_status must report it (and a healthy call after it must not)."""
    import ctypes as C
    from ssdr_al import _lib
    from ssdr_al._lib import DevArray
    n = 64
    p = (np.random.default_rng(2).random((n, 3)) * 0.01).astype(np.float32)          # one voxel at dl = 0.04
    f = np.zeros((n, 3), np.float32); lab = np.arange(n, dtype=np.int32).reshape(-1, 1)   # 64 distinct labels
    d_p, d_f, d_l = DevArray.from_host(p), DevArray.from_host(f), DevArray.from_host(lab)
    o_p, o_f, o_l, o_m = DevArray((n, 3), np.float32), DevArray((n, 3), np.float32), DevArray((n, 1), np.int32), DevArray((2,), np.int64)
    off = np.array([0, n], np.int64)
    L = _lib.lib()
    _lib.check(L.ssdr_grid_subsample_batch_dev(d_p.ptr, d_f.ptr, 3, d_l.ptr, 1, _lib.ptr(off), 1, 0.04, o_p.ptr, o_f.ptr, o_l.ptr, o_m.ptr, None))
    st = C.c_int32(0)
    assert L.ssdr_grid_subsample_status(None, C.byref(st)) != 0 and st.value & 1
    lab2 = (np.arange(n, dtype=np.int32) % 5).reshape(-1, 1)
    d_l2 = DevArray.from_host(lab2)
    _lib.check(L.ssdr_grid_subsample_batch_dev(d_p.ptr, d_f.ptr, 3, d_l2.ptr, 1, _lib.ptr(off), 1, 0.04, o_p.ptr, o_f.ptr, o_l.ptr, o_m.ptr, None))
    _lib.check(L.ssdr_grid_subsample_status(None, C.byref(st)))
    assert st.value == 0 and int(o_m.to_host()[0]) == 1


def _batch(clouds, dl, method=None):
    from ssdr_al import _lib
    from ssdr_al._lib import DevArray
    L = _lib.lib()
    sizes = [len(c[0]) for c in clouds]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    P = np.concatenate([c[0] for c in clouds]); Fe = np.concatenate([c[1] for c in clouds]); Lb = np.concatenate([c[2] for c in clouds])
    d_p, d_f, d_l = DevArray.from_host(P), DevArray.from_host(Fe), DevArray.from_host(Lb)
    o_p, o_f, o_l = DevArray(P.shape, np.float32), DevArray(Fe.shape, np.float32), DevArray(Lb.shape, np.int32)
    d_m = DevArray((len(sizes),), np.int64)
    if method is not None:
        _lib.check(L.ssdr_grid_subsample_set_method(method))
    try:
        _lib.check(L.ssdr_grid_subsample_batch_dev(d_p.ptr, d_f.ptr, Fe.shape[1], d_l.ptr, Lb.shape[1], _lib.ptr(off), len(sizes), dl, o_p.ptr, o_f.ptr, o_l.ptr, d_m.ptr, None))
        import ctypes
        st = ctypes.c_int32()
        rc = L.ssdr_grid_subsample_status(None, ctypes.byref(st))
    finally:
        _lib.check(L.ssdr_grid_subsample_set_method(0))
    m = d_m.to_host(); gp, gf, gl = o_p.to_host(), o_f.to_host(), o_l.to_host()
    return rc, st.value, [(gp[int(off[r]):int(off[r]) + int(m[r])], gf[int(off[r]):int(off[r]) + int(m[r])], gl[int(off[r]):int(off[r]) + int(m[r])]) for r in range(len(sizes))]


def _cloud(rng, n, box, shift=0.0, nlab=13):
    p = (rng.random((n, 3), dtype=np.float32) * np.asarray(box, np.float32) + np.float32(shift)).astype(np.float32)
    return p, rng.integers(0, 256, (n, 3)).astype(np.float32), rng.integers(0, nlab, (n, 1)).astype(np.int32)


def test_subsample_batch_partition_paths(backend, orc):
    """The bucket-partition implementation of the batch flavour (frontend.hip) on the shapes that leave its common case: a bucket with more
    records than its LDS holds (cut into slices of voxels, rows through the overflow region), the 16-voxel-wide bucket geometry of a long
    grid, negative coordinates, a plane — against the oracle, and against the sort-based implementation of the same entry point."""
    rng = np.random.default_rng(77)
    big = backend != "emu"
    clouds = [_cloud(rng, 9000 if not big else 60000, (0.33, 0.33, 0.33), shift=-0.2),      # ~512 voxels, 18+ points each: buckets of > 1536 records
              _cloud(rng, 3000 if not big else 400000, (21.0, 13.0, 3.0)),                   # 525 x 325 x 75 voxels: 16 x 8 x 8 buckets
              _cloud(rng, 2500 if not big else 300000, (5.0, 4.0, 0.0), shift=-3.0),          # a plane at z = -3
              _cloud(rng, 64, (0.01, 0.01, 0.01))]                                            # one voxel
    rc, st, got = _batch(clouds, 0.04)
    assert rc == 0 and st == 0
    rc2, st2, got_sort = _batch(clouds, 0.04, method=1)
    assert rc2 == 0 and st2 == 0
    for r, (p, f, l) in enumerate(clouds):
        exp = orc.grid_subsampling(p, f, l, 0.04, order="key")
        for x, y, z in zip(got[r], exp, got_sort[r]):
            assert_bits_equal(x, y, "cloud %d" % r); assert_bits_equal(z, y, "cloud %d (sort)" % r)


def test_subsample_batch_partition_reports_what_it_cannot_take(backend, orc):
    """A grid of more than 16384 buckets, and a voxel of more than 1024 points (FE_CAP: ~1100 and 2500 are both tried): reported by
    ssdr_grid_subsample_status; the sort-based implementation of the same entry point then gives the reference's rows."""
    rng = np.random.default_rng(78)
    wide = [_cloud(rng, 4000, (60.0, 40.0, 3.0)), _cloud(rng, 500, (1.0, 1.0, 1.0))]
    rc, st, _ = _batch(wide, 0.04)
    assert rc != 0 and st & 2
    dense = [_cloud(rng, 500, (1.0, 1.0, 1.0)), _cloud(rng, 2500, (0.01, 0.01, 0.01), shift=0.5)]
    rc, st, got = _batch(dense, 0.04)
    assert rc != 0 and st & 4
    exp0 = orc.grid_subsampling(*dense[0], 0.04, order="key")          # the other cloud of the call is not affected
    for x, y in zip(got[0], exp0):
        assert_bits_equal(x, y)
    just_over = [_cloud(rng, 1100, (0.01, 0.01, 0.01), shift=0.5), _cloud(rng, 500, (1.0, 1.0, 1.0))]      # a voxel of ~1100 points: past the 1024 the reduction holds
    rc, st, _ = _batch(just_over, 0.04)
    assert rc != 0 and st & 4
    for clouds in (wide, dense, just_over):
        rc, st, got = _batch(clouds, 0.04, method=1)
        assert rc == 0 and st == 0
        for r, (p, f, l) in enumerate(clouds):
            for x, y in zip(got[r], orc.grid_subsampling(p, f, l, 0.04, order="key")):
                assert_bits_equal(x, y)
