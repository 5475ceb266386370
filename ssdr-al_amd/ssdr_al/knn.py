"""Host-side mirror of the reference's ``nearest_neighbors`` module
(/root/reference/SSDR_AL_s3dis/utils/nearest_neighbors/knn.pyx:33-109): same names, argument meaning,
dtypes and return shapes; the work is done by libssdr_al.so's kd-tree kernels."""
import ctypes as C

import numpy as np

from . import _lib


def knn(pts, queries, K, omp=False):
    """knn.pyx:33-69 — pts [Np,3], queries [Nq,3] -> int64 [Nq,K].  ``omp`` is accepted and ignored
    (the reference uses it to pick the OpenMP variant; the result is the same)."""
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = pts_c if queries is pts else np.ascontiguousarray(queries, dtype=np.float32)
    indices = np.zeros((q_c.shape[0], K), dtype=np.int64)
    _lib.check(_lib.lib().ssdr_knn(_lib.ptr(pts_c), pts_c.shape[0], pts_c.shape[1], _lib.ptr(q_c), q_c.shape[0],
                                   int(K), _lib.ptr(indices)))
    return indices


def knn_batch(pts, queries, K, omp=False):
    """knn.pyx:71-109 — pts [B,Np,3], queries [B,Nq,3] -> int64 [B,Nq,K]."""
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = pts_c if queries is pts else np.ascontiguousarray(queries, dtype=np.float32)
    indices = np.zeros((pts_c.shape[0], q_c.shape[1], K), dtype=np.int64)
    _lib.check(_lib.lib().ssdr_knn_batch(_lib.ptr(pts_c), pts_c.shape[0], pts_c.shape[1], pts_c.shape[2],
                                         _lib.ptr(q_c), q_c.shape[1], int(K), _lib.ptr(indices)))
    return indices


def knn_batch_distance_pick(pts, nqueries, K, omp=False, seed=None):
    """knn.pyx:111-149 — pts [B,Np,3] -> (indices int64 [B,nqueries,K], queries float32 [B,nqueries,3]): query points picked
    "least used first".  The reference seeds std::mt19937 with time(0); `seed=None` does the same, an int reproduces a run
    (of the reference too, given the same clock value).  `omp` is accepted and ignored."""
    import time
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    indices = np.zeros((pts_c.shape[0], nqueries, K), dtype=np.int64)
    queries = np.zeros((pts_c.shape[0], nqueries, pts_c.shape[2]), dtype=np.float32)
    s = int(time.time()) if seed is None else int(seed)
    _lib.check(_lib.lib().ssdr_knn_batch_distance_pick(_lib.ptr(pts_c), pts_c.shape[0], pts_c.shape[1], pts_c.shape[2], _lib.ptr(queries),
                                                       int(nqueries), int(K), _lib.ptr(indices), s & 0xffffffff))
    return indices, queries


def knn_batch_i32(pts, queries, K):
    """Same search, int32 result (what DataProcessing.knn_search returns, helper_tool.py:182-183)."""
    pts_c = np.ascontiguousarray(pts, dtype=np.float32)
    q_c = pts_c if queries is pts else np.ascontiguousarray(queries, dtype=np.float32)
    indices = np.zeros((pts_c.shape[0], q_c.shape[1], K), dtype=np.int32)
    _lib.check(_lib.lib().ssdr_knn_batch_i32(_lib.ptr(pts_c), pts_c.shape[0], pts_c.shape[1], pts_c.shape[2],
                                             _lib.ptr(q_c), q_c.shape[1], int(K), _lib.ptr(indices)))
    return indices


def knn_pyramid(xyz, ratios, K):
    """The loop of tf_map (s3dis_dataset.py:164-177) in one call.

    xyz [B,N,3] -> (neigh_idx, sub_idx, interp_idx): lists over levels of int32 arrays
    [B,N_i,K], [B,N_{i+1},K], [B,N_i,1]."""
    xyz_c = np.ascontiguousarray(xyz, dtype=np.float32)
    B, N = xyz_c.shape[0], xyz_c.shape[1]
    L = len(ratios)
    sizes = [N]
    for r in ratios:
        sizes.append(sizes[-1] // int(r))
    neigh = [np.zeros((B, sizes[i], K), np.int32) for i in range(L)]
    sub = [np.zeros((B, sizes[i + 1], K), np.int32) for i in range(L)]
    interp = [np.zeros((B, sizes[i], 1), np.int32) for i in range(L)]
    arr = C.c_void_p * L
    r = np.asarray(ratios, dtype=np.int32)
    _lib.check(_lib.lib().ssdr_knn_pyramid(_lib.ptr(xyz_c), B, N, L, _lib.ptr(r), int(K),
                                           arr(*[a.ctypes.data for a in neigh]), arr(*[a.ctypes.data for a in sub]),
                                           arr(*[a.ctypes.data for a in interp])))
    return neigh, sub, interp


def knn_status(stream=None, wait=True):
    """Waits for `stream` and raises if the last device-flavour KNN call issued on it overflowed a device-side capacity
    (ssdr_knn_status).  wait=False (ssdr_knn_status_poll): only the calls that have finished are looked at — for a caller that keeps
    several batches in flight on the stream.  Returns (rows handed to the tree walk for K=16, for K=1, status bits, deepest tree)."""
    out = (C.c_int32 * 4)()
    L = _lib.lib()
    _lib.check(L.ssdr_knn_status(stream, out) if wait else L.ssdr_knn_status_poll(stream, out))
    return tuple(out)
