"""k-NN structures of the superpoint partition on the GPU, with the reference's call signatures
(`partition/graphs.py:7-70`).  Only the `voronoi == 0` branch exists (the Delaunay variant is not used by
`compute_superpoint.py:47`)."""
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
    if voronoi > 0:
        raise NotImplementedError("the Delaunay / voronoi branch of compute_graph_nn_2 is not built")
    return _knn_graph(xyz, k_nn1, k_nn2)
