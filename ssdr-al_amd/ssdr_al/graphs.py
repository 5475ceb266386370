"""k-NN structures of the superpoint partition on the GPU, with the reference's call signatures
(`partition/graphs.py:7-70`).  The k-NN searches run on the GPU (float64 kd walk, sklearn's semantics).  The `voronoi > 0`
branch (:38-62, not taken by `compute_superpoint.py:47`) triangulates with `scipy.spatial.Delaunay` exactly as the reference does —
Qhull on the host, there is no other arithmetic in it to move — and merges its short edges with the GPU's k-NN edges."""
import numpy as np

from . import _lib
from ._lib import DevArray


def _knn_graph(xyz, k_nn1, k_nn2):
    assert k_nn1 <= k_nn2, "knn1 must be smaller than knn2"
    xyz = np.ascontiguousarray(xyz, np.float32)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be [n,3]")
    n = xyz.shape[0]
    d_x = DevArray.from_host(xyz)
    src, tgt = DevArray((n * k_nn1,), np.uint32), DevArray((n * k_nn1,), np.uint32)
    dist, tgt2 = DevArray((n * k_nn1,), np.float32), DevArray((n * k_nn2,), np.uint32)
    _lib.check(_lib.lib().ssdr_knn_graph_dev(d_x.ptr, n, k_nn1, k_nn2, src.ptr, tgt.ptr, dist.ptr, tgt2.ptr, None))
    _lib.sync()
    return dict([("is_nn", True), ("source", src.to_host()), ("target", tgt.to_host()), ("distances", dist.to_host())]), tgt2.to_host()


def compute_graph_nn(xyz, k_nn):
    """graphs.py:7-22"""
    return _knn_graph(xyz, k_nn, k_nn)[0]


def compute_graph_nn_2(xyz, k_nn1, k_nn2, voronoi=0.0):
    """graphs.py:23-70: (graph of the k_nn1 neighbours, flat targets of the k_nn2 neighbours)"""
    graph, target2 = _knn_graph(xyz, k_nn1, k_nn2)
    if voronoi > 0:
        # graphs.py:38-62.  The six edges of every Delaunay tetrahedron (edge-major, as the reference stacks them: its `distances` keep that order),
        # those shorter than sqrt(voronoi) joined with the k_nn1 edges, one copy of every (source, target) pair in order of source + n * target
        from scipy.spatial import Delaunay
        tets = Delaunay(xyz).simplices.astype(np.uint64)          # (`tri.vertices` until SciPy 1.11)
        ends = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
        a, b = tets[:, ends[:, 0]].T.ravel(), tets[:, ends[:, 1]].T.ravel()
        d2 = np.square(xyz[a] - xyz[b]).sum(axis=1)
        short = d2 < voronoi
        src = np.concatenate([a[short], graph["source"]])
        tgt = np.concatenate([b[short], graph["target"]])
        first = np.unique(src + np.uint64(xyz.shape[0]) * tgt, return_index=True)[1]
        graph.update(source=src[first], target=tgt[first], distances=d2[short])      # (:60 leaves the Delaunay edges' squared lengths only)
    return graph, target2
