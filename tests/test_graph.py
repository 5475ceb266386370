"""Superpoint-graph inputs: k-NN structures against the reference's own partition/graphs.py (golden vectors
made by importing it, tests/golden/make_golden_graph.py) and against the oracle; geometric features against the oracle
(parity unpinned for those: the reference's ply_c.cpp needs Eigen + Boost, see oracle/graph_np.py)."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gg():
    return np.load(os.path.join(HERE, "golden", "graph_golden.npz"))


def test_oracle_graph_matches_reference_golden(gg):
    from oracle import graph_np
    for name in ("uniform", "room"):
        g, t2 = graph_np.compute_graph_nn_2(gg[name + "/xyz"], 10, 45)
        for k in ("source", "target", "distances"):
            assert_bits_equal(g[k], gg[name + "/" + k], name + " " + k)
        assert_bits_equal(t2, gg[name + "/target2"], name + " target2")


def test_knn_graph_matches_reference_golden(backend, gg):
    from ssdr_al import graphs
    for name in ("uniform", "room"):
        x = gg[name + "/xyz"]
        g, t2 = graphs.compute_graph_nn_2(x, 10, 45)
        assert g["is_nn"] is True and g["source"].dtype == np.uint32 and g["distances"].dtype == np.float32 and t2.dtype == np.uint32
        for k in ("source", "target", "distances"):
            assert_bits_equal(g[k], gg[name + "/" + k], name + " " + k)
        assert_bits_equal(t2, gg[name + "/target2"], name + " target2")
        g1 = graphs.compute_graph_nn(x, 7)
        assert_bits_equal(g1["target"], gg[name + "/nn7_target"]); assert_bits_equal(g1["distances"], gg[name + "/nn7_distances"])
    with pytest.raises(AssertionError, match="knn1 must be smaller than knn2"):
        graphs.compute_graph_nn_2(gg["uniform/xyz"], 12, 10)
    # Delaunay branch (graphs.py:38-62): Qhull on the host as in the reference, k-NN edges from the GPU
    g, t2 = graphs.compute_graph_nn_2(gg["uniform/xyz"][:300], 5, 12, voronoi=0.3)
    for k in ("source", "target", "distances"):
        assert g[k].dtype == gg["voronoi/" + k].dtype and np.array_equal(g[k], gg["voronoi/" + k]), k
    assert_bits_equal(t2, gg["voronoi/target2"], "voronoi target2")


def test_knn_graph_fresh_inputs_against_oracle(backend):
    from oracle import graph_np
    from ssdr_al import graphs
    rng = np.random.default_rng(5)
    n = 700 if backend == "emu" else 6000
    x = (rng.normal(0, 1, (n, 3)) * np.array([3, 2, 0.05])).astype(np.float32)          # a slab: long kd-tree walks
    g, t2 = graphs.compute_graph_nn_2(x, 10, 45)
    eg, et2 = graph_np.compute_graph_nn_2(x, 10, 45)
    assert_bits_equal(t2, et2); assert_bits_equal(g["target"], eg["target"]); assert_bits_equal(g["distances"], eg["distances"])


def test_geof_against_oracle(backend, gg):
    from oracle import graph_np
    from ssdr_al import libply_c
    x = gg["room/xyz"]
    t2 = gg["room/target2"]
    got = libply_c.compute_geof(x, t2, 45)
    exp = graph_np.compute_geof(x, t2, 45)
    assert got.shape == (len(x), 4) and got.dtype == np.float32
    # float32 covariance + float64 eigen solve on both sides; the summation order of the covariance differs (1e-4)
    assert np.abs(got[:, :3] - exp[:, :3]).max() < 2e-4
    assert np.abs(got[:, 3] - exp[:, 3]).max() < 2e-3            # verticality mixes eigenvectors: looser where two eigenvalues are close
    assert (got[:, :3] >= -1e-6).all() and (got[:, 0] + got[:, 1] + got[:, 2] < 1 + 1e-4).all()     # linearity + planarity + scattering = 1
    # a plane: planarity ~ 1, its normal is vertical -> verticality ~ 0 (the weighted |eigenvector| sum lies in the plane)
    rng = np.random.default_rng(2)
    p = np.concatenate([rng.random((400, 2)), np.zeros((400, 1))], 1).astype(np.float32)
    from oracle.graph_np import compute_graph_nn_2
    _, tp = compute_graph_nn_2(p, 5, 20)
    gp = libply_c.compute_geof(p, tp, 20)
    assert (gp[:, 2] < 1e-3).all() and (gp[:, 3] < 1e-3).all()


def test_prune_against_restatement(backend):
    """libply_c.prune (ply_c.cpp:289-383): PARITY UNPINNED by the reference (Boost.Python / Eigen absent); bit-exact against the loop
    restatement in oracle/graph_np.py, call forms of partition/partition.py:126-144."""
    from oracle import graph_np as G
    from ssdr_al import libply_c
    rng = np.random.default_rng(4)
    n = 3000 if backend == "emu" else 400000
    xyz = (rng.random((n, 3)) * np.array([6, 4, 3])).astype(np.float32)
    xyz[: n // 3, 2] = 0.5                                              # a plane: several points per voxel
    xyz[n // 2] = xyz.max(0)                                            # a point on the upper faces of the box
    rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
    lab = rng.integers(0, 14, n).astype(np.uint8); obj = rng.integers(0, 40, n).astype(np.uint32)
    for args in ((0.05 if backend == "gpu" else 0.3, rgb, lab, np.zeros(1, np.uint8), 13, 0), (0.5, rgb, lab, obj, 13, 39),
                 (0.4, np.zeros(xyz.shape, np.uint8), np.array(1, np.uint8), np.zeros(1, np.uint8), 0, 0)):
        got = libply_c.prune(xyz, *args)
        exp = G.prune(xyz, *args)
        assert got[0].shape == exp[0].shape and got[0].shape[0] < n
        for a, b, what in zip(got, exp, ("xyz", "rgb", "labels", "objects")):
            if what == "labels" and args[4] == 0 or what == "objects" and args[5] == 0:
                assert a.shape == (len(exp[0]), 1) and not a.any()
                continue
            assert a.dtype == b.dtype and np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), what
        assert int(got[2].sum()) == n if args[4] else True
