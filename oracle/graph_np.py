"""TEST INFRASTRUCTURE ONLY (parity checker): CPU restatement of the superpoint-graph inputs.

compute_graph_nn_2 follows partition/graphs.py:23-70 (voronoi == 0) with sklearn's arithmetic spelled out (float64
coordinates, rdist = ((dx*dx + dy*dy) + dz*dz), neighbours by ascending distance, sqrt); pinned against the reference's
own function (tests/golden/make_golden_graph.py imports it) in tests/test_graph.py.

compute_geof follows partition/ply_c/ply_c.cpp:385-455.  PARITY UNPINNED: that file needs Eigen and Boost.Python, absent
from the build image; the reference's float32 EigenSolver is replaced by numpy's float64 symmetric solver on the float32
covariance, so agreement with the reference itself is expected only to a few 1e-4 where eigenvalues are well separated.
"""
import numpy as np


def knn_f64(xyz, k):
    X = np.asarray(xyz, np.float32).astype(np.float64)
    d2 = np.zeros((len(X), len(X)))
    for c in range(3):
        diff = X[:, None, c] - X[None, :, c]
        d2 = d2 + diff * diff
    idx = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return idx, np.sqrt(np.take_along_axis(d2, idx, 1))


def compute_graph_nn_2(xyz, k_nn1, k_nn2):
    n = len(xyz)
    neighbors, distances = knn_f64(xyz, k_nn2 + 1)
    neighbors, distances = neighbors[:, 1:], distances[:, 1:]
    graph = {"is_nn": True,
             "source": np.repeat(np.arange(n), k_nn1).astype(np.uint32),
             "target": neighbors[:, :k_nn1].reshape(-1).astype(np.uint32),
             "distances": distances[:, :k_nn1].reshape(-1).astype(np.float32)}
    return graph, neighbors.reshape(-1).astype(np.uint32)


def compute_geof(xyz, target, k_nn):
    xyz = np.asarray(xyz, np.float32)
    n = len(xyz)
    nb = np.asarray(target).reshape(n, k_nn).astype(np.int64)
    pos = np.concatenate([xyz[:, None, :], xyz[nb]], 1)                                   # [n, k+1, 3] float32
    cen = pos - pos.mean(1, dtype=np.float32, keepdims=True)
    cov = np.einsum("nki,nkj->nij", cen, cen).astype(np.float32) / np.float32(k_nn + 1)
    lam, vec = np.linalg.eigh(cov.astype(np.float64))                                     # ascending
    lam, vec = lam[:, ::-1], vec[:, :, ::-1]
    lam = np.maximum(lam.astype(np.float32), 0)
    s = np.sqrt(lam)
    lin, pla, sca = (s[:, 0] - s[:, 1]) / s[:, 0], (s[:, 1] - s[:, 2]) / s[:, 0], s[:, 2] / s[:, 0]
    u = np.einsum("nk,ndk->nd", lam, np.abs(vec).astype(np.float32))
    ver = u[:, 2] / np.sqrt((u * u).sum(1))
    return np.stack([lin, pla, sca, ver], 1).astype(np.float32)


def prune(xyz, voxel_size, rgb, labels, objects, n_labels, n_objects):
    """libply_c.prune (partition/ply_c/ply_c.cpp:289-383 with AttributeGrid :160-287), restated with plain loops.
    PARITY UNPINNED (the file needs Boost.Python / Eigen; the reference holds no fixture for it): bin = floor((p - bbox_min) / w) in
    float32; voxels are numbered in the order in which they are first met (add_occurence :172-181); float32 position sums and uint32
    colour sums in input order; position = sum / (float)count, colour = (uint8)((float)sum / count); label / object histograms with
    n + 1 columns.  Returns (xyz f4 [m,3], rgb u1 [m,3], labels u4 [m,n_labels+1], objects u4 [m,n_objects+1])."""
    xyz = np.ascontiguousarray(xyz, np.float32); rgb = np.ascontiguousarray(rgb, np.uint8).reshape(len(xyz), 3)
    w = np.float32(voxel_size)
    mn = xyz.min(0)
    bins = np.floor((xyz - mn) / w).astype(np.uint32)                  # :337-339, float32 arithmetic
    index = {}
    for i in range(len(xyz)):
        index.setdefault((int(bins[i, 0]), int(bins[i, 1]), int(bins[i, 2])), len(index))
    m = len(index)
    cnt = np.zeros(m, np.uint32); acc = np.zeros((m, 3), np.float32); col = np.zeros((m, 3), np.uint32)
    hl = np.zeros((m, n_labels + 1), np.uint32); ho = np.zeros((m, n_objects + 1), np.uint32)
    for i in range(len(xyz)):
        v = index[(int(bins[i, 0]), int(bins[i, 1]), int(bins[i, 2]))]
        cnt[v] += 1
        acc[v] = acc[v] + xyz[i]                                       # float32 adds, input order
        col[v] += rgb[i]
        if n_labels > 0:
            hl[v, int(labels[i])] += 1
        if n_objects > 0:
            ho[v, int(objects[i])] += 1
    c = cnt.astype(np.float32)[:, None]
    return acc / c, (col.astype(np.float32) / c).astype(np.uint8), hl, ho
