"""KNN parity: the HIP kd-tree path through the C ABI against the golden vectors of the reference build and
against the oracle on fresh inputs.  Bit-exact index equality, ties included."""
import os
import numpy as np
import pytest

from conftest import assert_bits_equal

KNN_CASES = ["uniform", "room", "lattice", "duplicate_padded", "tiny7", "all_same"]


@pytest.mark.parametrize("case", KNN_CASES)
def test_knn_matches_reference_golden(backend, golden, case):
    from ssdr_al import knn
    g = golden("knn_golden.npz")
    p = g[case + "/pts"]
    sub = p[: max(1, len(p) // 4)]
    assert_bits_equal(knn.knn(p, p, 16).astype(np.int32), g[case + "/self16"], "self16")
    assert_bits_equal(knn.knn(p, p, 1).astype(np.int32), g[case + "/self1"], "self1")
    assert_bits_equal(knn.knn(p, p, 5).astype(np.int32), g[case + "/self5"], "self5 (generic-K kernel)")
    assert_bits_equal(knn.knn(sub, p, 1).astype(np.int32), g[case + "/up1"], "up1")


def test_knn_batch_matches_reference_golden(backend, golden):
    from ssdr_al import knn
    g = golden("knn_golden.npz")
    out = knn.knn_batch(g["batch/pts"], g["batch/q"], 16, omp=True)
    assert out.dtype == np.int64 and out.shape == (3, 150, 16)
    assert_bits_equal(out.astype(np.int32), g["batch/idx16"])
    assert_bits_equal(knn.knn_batch_i32(g["batch/pts"], g["batch/q"], 16), g["batch/idx16"])


def test_pyramid_matches_reference_golden(backend, golden):
    from ssdr_al import knn
    g = golden("pyramid_golden.npz")
    neigh, sub, interp = knn.knn_pyramid(g["xyz"], g["ratios"], 16)
    for i in range(5):
        assert_bits_equal(neigh[i], g["neigh%d" % i], "neigh%d" % i)
        assert_bits_equal(sub[i], g["sub%d" % i], "sub%d" % i)
        assert_bits_equal(interp[i], g["interp%d" % i], "interp%d" % i)


def test_knn_k_larger_than_support(backend, orc):
    from ssdr_al import knn
    rng = np.random.default_rng(3)
    p = rng.random((7, 3), dtype=np.float32)
    q = rng.random((20, 3), dtype=np.float32)
    out = knn.knn(p, q, 16)
    assert_bits_equal(out, orc.knn(p, q, 16))
    assert (out[:, 7:] == 0).all()          # knn_.cxx:30-31: zero-initialised, never written


def test_device_path_reports_tree_depth_overflow(backend):
    """A support set whose exact nanoflann tree is deeper than the level limit (geometric spacing: every split peels one point off)
    with duplicated points (tie rows -> the tree is needed): the device flavour only enqueues, ssdr_knn_status must report it."""
    import ctypes as C
    from ssdr_al import _lib, knn
    from ssdr_al._lib import DevArray
    x = np.repeat((2.0 ** -np.arange(110)).astype(np.float32), 2)
    p = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)[None]
    d_p = DevArray.from_host(p); d_o = DevArray((1, p.shape[1], 16), np.int32)
    _lib.check(_lib.lib().ssdr_knn_batch_dev(d_p.ptr, 1, p.shape[1], 3, d_p.ptr, p.shape[1], 16, d_o.ptr, None))
    with pytest.raises(_lib.SsdrError) as e:
        knn.knn_status()
    assert "0x4" in str(e.value) or "status" in str(e.value)
    # the non-waiting flavour (ssdr_knn_status_poll) reports the same through the call's ticket once the call has finished
    _lib.check(_lib.lib().ssdr_knn_batch_dev(d_p.ptr, 1, p.shape[1], 3, d_p.ptr, p.shape[1], 16, d_o.ptr, None))
    _lib.sync()
    with pytest.raises(_lib.SsdrError):
        knn.knn_status(wait=False)
    assert knn.knn_status(wait=False)[2] == 0            # reported once, then clear
    # a healthy call on the same stream clears it
    q = np.random.default_rng(0).random((1, 500, 3), dtype=np.float32)
    d_q = DevArray.from_host(q); d_o2 = DevArray((1, 500, 16), np.int32)
    _lib.check(_lib.lib().ssdr_knn_batch_dev(d_q.ptr, 1, 500, 3, d_q.ptr, 500, 16, d_o2.ptr, None))
    assert knn.knn_status()[2] == 0


def test_blocking_status_keeps_an_earlier_calls_error(backend):
    """bad call, then a good call on the same stream: the blocking ssdr_knn_status reads only the newest call's counters directly, so
    it must fold the earlier call's ticket instead of dropping it (batches in flight: the host asks once per step, not once per call)"""
    from ssdr_al import _lib, knn
    from ssdr_al._lib import DevArray
    x = np.repeat((2.0 ** -np.arange(110)).astype(np.float32), 2)
    p = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)[None]
    d_p = DevArray.from_host(p); d_o = DevArray((1, p.shape[1], 16), np.int32)
    q = np.random.default_rng(0).random((1, 500, 3), dtype=np.float32)
    d_q = DevArray.from_host(q); d_o2 = DevArray((1, 500, 16), np.int32)
    _lib.check(_lib.lib().ssdr_knn_batch_dev(d_p.ptr, 1, p.shape[1], 3, d_p.ptr, p.shape[1], 16, d_o.ptr, None))      # tree too deep
    _lib.check(_lib.lib().ssdr_knn_batch_dev(d_q.ptr, 1, 500, 3, d_q.ptr, 500, 16, d_o2.ptr, None))                  # healthy
    with pytest.raises(_lib.SsdrError):
        knn.knn_status(wait=True)
    assert knn.knn_status(wait=True)[2] == 0             # reported once


def test_status_tickets_wrap_around(backend, orc):
    """more device-flavour calls on a stream than it has status tickets (16) between two polls: the oldest are folded, nothing is lost or stuck"""
    from ssdr_al import _lib, knn
    from ssdr_al._lib import DevArray
    q = np.random.default_rng(3).random((1, 300, 3), dtype=np.float32)
    d_q = DevArray.from_host(q); d_o = DevArray((1, 300, 16), np.int32)
    for _ in range(40):
        _lib.check(_lib.lib().ssdr_knn_batch_dev(d_q.ptr, 1, 300, 3, d_q.ptr, 300, 16, d_o.ptr, None))
    _lib.sync()
    assert knn.knn_status(wait=False)[2] == 0
    assert knn.knn_status()[2] == 0
    assert_bits_equal(d_o.to_host(), orc.knn_batch(q, q, 16, threads=2))


def test_generic_k_up_to_256(backend, orc):
    """K = 200 goes through the LDS result set (128 KiB of dynamic LDS: needs the opt-in above 64 KiB)."""
    from ssdr_al import knn
    rng = np.random.default_rng(12)
    p = rng.random((700, 3), dtype=np.float32)
    assert_bits_equal(knn.knn(p, p, 200), orc.knn(p, p, 200), "K=200")
    idx, qs = knn.knn_batch_distance_pick(p[None], 6, 128, seed=3)       # 64 KiB dynamic + static LDS
    assert idx.shape == (1, 6, 128)


def test_knn_rejects_unsupported_dim(backend):
    from ssdr_al import _lib, knn
    with pytest.raises(_lib.SsdrError) as e:
        knn.knn(np.zeros((8, 4), np.float32), np.zeros((8, 4), np.float32), 2)
    assert e.value.status == 5


@pytest.mark.parametrize("dim", [1, 2])
def test_knn_one_and_two_dimensions(backend, dim):
    """knn_.cxx:22-135 is dim-generic; dim 1 and 2 are answered as the 3-D problem with zero coordinates, bit for bit the REAL reference's answer
    (the compiled reference in the build container, the dim-generic C oracle elsewhere): lattices and duplicates for the tie order."""
    import oracle
    from ssdr_al import knn
    rng = np.random.default_rng(31 + dim)
    n = 1500 if backend == "emu" else 20000
    pts = rng.random((n, dim)).astype(np.float32)
    pts[: n // 4] = np.round(pts[: n // 4] * 20) / 20            # a lattice: equal distances
    pts[-50:] = pts[:50]                                         # duplicates
    q = np.concatenate([pts[:200], rng.random((100, dim)).astype(np.float32)])
    r = oracle.ref()
    exp = r.knn(pts, q, 16, omp=False) if r is not None else oracle.c().knn(pts, q, 16)
    assert_bits_equal(knn.knn(pts, q, 16), exp, "dim %d" % dim)
    assert_bits_equal(knn.knn_batch(pts[None], pts[None], 1)[0], (r.knn(pts, pts, 1) if r is not None else oracle.c().knn(pts, pts, 1)), "dim %d, K = 1" % dim)


def test_knn_fresh_inputs_against_oracle(backend, orc):
    from ssdr_al import knn
    rng = np.random.default_rng(11)
    n = 3000 if backend == "emu" else 40960
    p = (rng.random((n, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    p[: n // 3, 2] = 0
    p[-n // 5:] = p[: n // 5]                 # padded-by-duplication tile (s3dis_dataset.py:147-150)
    p = p[rng.permutation(n)][None]
    assert_bits_equal(knn.knn_batch(p, p, 16), orc.knn_batch(p, p, 16, threads=4))
    assert_bits_equal(knn.knn_batch(p[:, : n // 4], p, 1), orc.knn_batch(p[:, : n // 4], p, 1, threads=4))


@pytest.mark.parametrize("scale", ["1e-9", "0.25", "0"])
def test_tree_hand_over_cut_to_balls(backend, orc, scale, monkeypatch, capfd):
    """The hand-over trees are split only where the balls of the handed-over rows reach (SSDR_KNN_BALL_SCALE x the (K+1)-th squared
    distance; 0 = complete trees).  Balls far too small make the walks leave them: those rows must come back through the fall-back
    list and the complete trees with the reference's answer."""
    from ssdr_al import knn
    rng = np.random.default_rng(21)
    n = 4000 if backend == "emu" else 40960
    p = (rng.random((n, 3), dtype=np.float32) * np.array([8, 6, 3], np.float32)).astype(np.float32)
    p[-40:] = p[:40]                          # a few duplicated points: tie rows around them, answered by the tree (fewer than the ball list holds)
    p = p[rng.permutation(n)][None]
    monkeypatch.setenv("SSDR_KNN_BALL_SCALE", scale)
    monkeypatch.setenv("SSDR_KNN_DEBUG", "1")
    got16 = knn.knn_batch(p, p, 16)
    st = knn.knn_status()
    err = capfd.readouterr().err
    got1 = knn.knn_batch(p[:, : n // 4], p, 1)
    assert 0 < st[0] < 2048, "the case must hand over some rows, fewer than the ball list holds (%d)" % st[0]
    again = int(err.split(";")[-1].split()[0]) if "again on complete trees" in err else -1
    if scale == "1e-9":
        assert again > 0, err
    if scale == "0":
        assert again == 0, err
    assert_bits_equal(got16, orc.knn_batch(p, p, 16, threads=4), "K=16 scale " + scale)
    assert_bits_equal(got1, orc.knn_batch(p[:, : n // 4], p, 1, threads=4), "K=1 scale " + scale)


@pytest.mark.gpu
def test_complete_rebuild_of_several_large_trees_in_one_launch(monkeypatch, capfd):
    """The re-build behind the hand-over (complete trees for rows the ball-cut trees could not settle) takes one workgroup per flagged tree up to 4096 points
    and kd_levels_kernel — W co-operating workgroups, a meeting point per level — above: a 40 960-point tree took its one workgroup 6.6 ms.  Balls far too
    small (SSDR_KNN_BALL_SCALE=1e-9) send every handed-over row of a four-tile pyramid there: several large trees at once, levels 0 and 1 through the new
    kernel, the deeper ones through the old one; the pyramid must be the default path's, bit for bit, and SSDR_KD_LEVELS=0 (the old kernel alone) as well."""
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    from ssdr_al import _lib, knn
    _lib.use(GPU_LIB)
    try:
        rng = np.random.default_rng(77)
        B, N = 4, 40960
        xyz = (rng.random((B, N, 3), dtype=np.float32) * np.array([9, 7, 3], np.float32)).astype(np.float32)
        for b in range(B):                                 # a few duplicated points per tile: tie rows around them, fewer than the ball list holds
            xyz[b, -6 - 2 * b:] = xyz[b, : 6 + 2 * b]
            xyz[b] = xyz[b][rng.permutation(N)]
        ref = knn.knn_pyramid(xyz, [4, 4, 4, 4, 2], 16)
        monkeypatch.setenv("SSDR_KNN_BALL_SCALE", "1e-9")
        monkeypatch.setenv("SSDR_KNN_DEBUG", "1")
        capfd.readouterr()
        got = knn.knn_pyramid(xyz, [4, 4, 4, 4, 2], 16)
        assert knn.knn_status()[0] > 0                     # rows were handed over (the status call prints the debug line)
        err = capfd.readouterr().err
        line = [ln for ln in err.splitlines() if "again on complete trees" in ln][-1]
        assert int(line.split(";")[-1].split()[0]) > 20, line      # the K = 16 rows that went through the complete trees
        for a, b in zip(ref, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y)
    finally:
        _lib.use(None)


@pytest.mark.parametrize("margin", ["-20", "0", "99"])
def test_tree_lower_levels_in_one_launch(backend, orc, margin, monkeypatch):
    """The kd forest's big nodes are split by level-wide launches down to the depth a balanced tree needs + SSDR_KD_REST_MARGIN levels, the rest by one
    launch whose workgroups walk their subtrees depth first (kd_rest_kernel).  -20: that launch builds everything from the roots; 99: it never runs
    (rounds 1-4).  Complete trees (ball scale 0) over clustered points with duplicates: deep, uneven trees, every tie row through the tree."""
    from ssdr_al import knn
    rng = np.random.default_rng(23)
    n = 3000 if backend == "emu" else 30000
    seats = rng.random((12, 3)) * np.array([8, 6, 3])
    p = (seats[rng.integers(0, 12, n)] + rng.normal(0, 1, (n, 3)) * np.exp(rng.normal(-2, 0.7, (n, 1)))).astype(np.float32)
    p[-60:] = p[:60]
    p = p[rng.permutation(n)][None]
    monkeypatch.setenv("SSDR_KNN_BALL_SCALE", "0")
    monkeypatch.setenv("SSDR_KD_REST_MARGIN", margin)
    got = knn.knn_batch(p, p, 16)
    st = knn.knn_status()
    assert st[0] > 0, "the case must hand rows over to the trees"
    assert_bits_equal(got, orc.knn_batch(p, p, 16, threads=4), "margin " + margin)


@pytest.mark.gpu
def test_pyramid_full_size_properties():
    """BASELINE config 2 shape: B=16 tiles of 40960 points.  Size-independent properties + oracle on 2 tiles."""
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    import oracle
    from ssdr_al import _lib, knn
    _lib.use(GPU_LIB)
    try:
        rng = np.random.default_rng(5)
        B, N = 16, 40960
        xyz = (rng.random((B, N, 3), dtype=np.float32) * np.array([10, 8, 3], np.float32)).astype(np.float32)
        xyz[:, : N // 2, 2] = 0
        xyz[3, -5000:] = xyz[3, :5000]
        neigh, sub, interp = knn.knn_pyramid(xyz, [4, 4, 4, 4, 2], 16)
        sizes = [40960, 10240, 2560, 640, 160, 80]
        for i in range(5):
            assert neigh[i].shape == (B, sizes[i], 16) and interp[i].shape == (B, sizes[i], 1)
            assert neigh[i].min() >= 0 and neigh[i].max() < sizes[i]
            assert interp[i].min() >= 0 and interp[i].max() < sizes[i + 1]
            assert np.array_equal(sub[i], neigh[i][:, : sizes[i + 1]])
            # ascending distances; the query itself is at distance 0
            pts = xyz[:, : sizes[i]]
            d = ((pts[:, :, None, :] - np.take_along_axis(pts[:, None], neigh[i][..., None].astype(np.int64), 2)
                  if False else 0))
            b = 3
            nb = pts[b][neigh[i][b]]
            dist = ((pts[b][:, None, :] - nb) ** 2).sum(-1)
            assert (np.diff(dist, axis=1) >= -1e-6).all()
            assert (dist[:, 0] == 0).all()
        o = oracle.c()
        for b in (0, 3):
            cur = xyz[b:b + 1]
            for i, r in enumerate([4, 4, 4, 4, 2]):
                assert_bits_equal(neigh[i][b:b + 1], o.knn_batch(cur, cur, 16).astype(np.int32), "tile %d level %d" % (b, i))
                s = cur[:, : cur.shape[1] // r]
                assert_bits_equal(interp[i][b:b + 1], o.knn_batch(s, cur, 1).astype(np.int32))
                cur = s
    finally:
        _lib.use(None)


def test_knn_batch_distance_pick_matches_reference(backend, orc):
    """knn_batch_distance_pick (knn.pyx:111-149): golden vectors from the REAL reference with its mt19937 clock seed pinned
    (tests/golden/make_golden.py: distance_pick_golden), the oracle, and the HIP path with the same seed."""
    import nearest_neighbors.lib.python.nearest_neighbors as nn
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distance_pick_golden.npz"))
    pts = g["dp/pts"]
    for seed, nq, K in ((1700000000, 120, 16), (7, 64, 1)):
        oi, oq = orc.knn_batch_distance_pick(pts, nq, K, seed)
        assert_bits_equal(oi, g["dp/%d/idx" % seed], "oracle idx"); assert_bits_equal(oq, g["dp/%d/q" % seed], "oracle queries")
        idx, q = nn.knn_batch_distance_pick(pts, nq, K, omp=True, seed=seed)
        assert idx.dtype == np.int64 and idx.shape == (3, nq, K) and q.dtype == np.float32 and q.shape == (3, nq, 3)
        assert_bits_equal(idx, g["dp/%d/idx" % seed], "idx seed %d" % seed)
        assert_bits_equal(q, g["dp/%d/q" % seed], "queries seed %d" % seed)
    # enough queries to exhaust the zero-count points: the "continue from the minimum count" branch (knn_.cxx:160-162)
    small = pts[:1, :40]
    idx, q = nn.knn_batch_distance_pick(small, 60, 4, seed=3)
    oi, oq = orc.knn_batch_distance_pick(small, 60, 4, 3)
    assert_bits_equal(idx, oi); assert_bits_equal(q, oq)
    # time-seeded like the reference when no seed is given: still a valid pick (every query is one of the points)
    idx, q = nn.knn_batch_distance_pick(small, 5, 4)
    assert all((small[0] == q[0, i]).all(1).any() for i in range(5))
