"""The oracle (oracle/*.c) against the golden vectors made from the real reference build, and against that
build itself where it exists (this container).  CPU only."""
import numpy as np
import pytest

from conftest import assert_bits_equal

KNN_CASES = ["uniform", "room", "lattice", "duplicate_padded", "tiny7", "all_same"]


@pytest.mark.parametrize("case", KNN_CASES)
def test_oracle_knn_matches_golden(orc, golden, case):
    g = golden("knn_golden.npz")
    p = g[case + "/pts"]
    sub = p[: max(1, len(p) // 4)]
    assert_bits_equal(orc.knn(p, p, 16).astype(np.int32), g[case + "/self16"], "self16")
    assert_bits_equal(orc.knn(p, p, 1).astype(np.int32), g[case + "/self1"], "self1")
    assert_bits_equal(orc.knn(p, p, 5).astype(np.int32), g[case + "/self5"], "self5")
    assert_bits_equal(orc.knn(sub, p, 1).astype(np.int32), g[case + "/up1"], "up1")


def test_oracle_knn_batch_matches_golden(orc, golden):
    g = golden("knn_golden.npz")
    assert_bits_equal(orc.knn_batch(g["batch/pts"], g["batch/q"], 16, threads=2).astype(np.int32), g["batch/idx16"])


def test_oracle_pyramid_matches_golden(orc, golden):
    g = golden("pyramid_golden.npz")
    cur = g["xyz"]
    for i, r in enumerate(g["ratios"]):
        neigh = orc.knn_batch(cur, cur, 16).astype(np.int32)
        sub = cur[:, : cur.shape[1] // r]
        assert_bits_equal(neigh, g["neigh%d" % i])
        assert_bits_equal(orc.knn_batch(sub, cur, 1).astype(np.int32), g["interp%d" % i])
        cur = sub


def test_oracle_subsample_matches_golden(orc, golden):
    g = golden("subsample_golden.npz")
    assert_bits_equal(orc.grid_subsampling(g["hand/pts"], sampleDl=1.0)[0], g["hand/out"], "hand order")
    for nm in ("tieA", "tieB"):
        assert_bits_equal(orc.grid_subsampling(g[nm + "/pts"], None, g[nm + "/cls"], 1.0)[1], g[nm + "/out_cls"], nm)
    p, f, c = orc.grid_subsampling(g["room/pts"], g["room/col"], g["room/lab"], 0.04)
    assert_bits_equal(p, g["room/out_pts"]); assert_bits_equal(f, g["room/out_col"]); assert_bits_equal(c, g["room/out_lab"])
    assert_bits_equal(orc.grid_subsampling(g["one/pts"], sampleDl=0.1)[0], g["one/out"])
    assert_bits_equal(orc.grid_subsampling(g["neg/pts"], sampleDl=0.3)[0], g["neg/out"])
    p, c = orc.grid_subsampling(g["manylab/pts"], None, g["manylab/cls"], 0.5)
    assert_bits_equal(p, g["manylab/out_pts"]); assert_bits_equal(c, g["manylab/out_cls"])


def test_oracle_sorted_order_is_a_permutation_of_reference_order(orc, golden):
    g = golden("subsample_golden.npz")
    a = orc.grid_subsampling(g["room/pts"], g["room/col"], g["room/lab"], 0.04, order="reference", return_keys=True)
    b = orc.grid_subsampling(g["room/pts"], g["room/col"], g["room/lab"], 0.04, order="key", return_keys=True)
    ia, ib = np.argsort(a[3]), np.argsort(b[3])
    assert (np.diff(b[3].astype(np.int64)) > 0).all()
    for x, y in zip(a[:3], b[:3]):
        assert_bits_equal(x[ia], y[ib])


def test_oracle_against_live_reference():
    """Fresh random inputs through the reference's own compiled C++ (only where /root/reference exists)."""
    import oracle
    ref = oracle.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    o = oracle.c()
    rng = np.random.default_rng(7)
    for n in (50, 1000, 3000):
        p = rng.random((n, 3), dtype=np.float32)
        p[n // 2:] = p[: n - n // 2]          # heavy duplication
        p = p[rng.permutation(n)]
        assert_bits_equal(o.knn(p, p, 16), ref.knn(p, p, 16))
        assert_bits_equal(o.knn(p[: n // 4], p, 1), ref.knn(p[: n // 4], p, 1))
    pts = rng.random((20000, 3), dtype=np.float32) * np.array([5, 4, 3], np.float32) - 2
    col = rng.integers(0, 256, (20000, 3)).astype(np.float32)
    lab = rng.integers(0, 13, (20000, 1)).astype(np.int32)
    for x, y in zip(o.grid_subsampling(pts, col, lab, 0.06), ref.grid_subsampling(pts, col, lab, 0.06)):
        assert_bits_equal(x, y)
