"""TEST INFRASTRUCTURE ONLY — ctypes access to the CPU oracle and (when built) the real reference ops.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker / the reported baseline.  The product (``ssdr-al_amd/``) never does.

``oracle.c``   -> liboracle.so            our C restatement (oracle/*.c), parity-pinned against
``oracle.ref`` -> _ref/libssdr_ref.so     the reference's own C++ compiled from /root/reference
                                          (oracle/Makefile target ``ref``; absent => ``ref is None``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle.so (always) and _ref/libssdr_ref.so (only where /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    ref = os.path.join(_HERE, "_ref", "libssdr_ref.so")
    if os.path.isdir("/root/reference") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(os.environ.get("SSDR_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so"))      # SSDR_ORACLE_LIB: the sanitizer build (tests/test_sanitizers.py)
        L = self.lib
        L.oracle_grid_subsampling.restype = C.c_long
        L.oracle_grid_subsampling.argtypes = [_f32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                              C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_knn.restype = None
        L.oracle_knn.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, _i64p, C.c_void_p]
        L.oracle_knn_batch.restype = None
        L.oracle_knn_batch.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t,
                                       _i64p, C.c_int]
        L.oracle_kdtree_dump.restype = C.c_long
        L.oracle_kdtree_dump.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _i64p, C.c_void_p, C.c_void_p,
                                         C.c_size_t]

    def grid_subsampling(self, points, features=None, classes=None, sampleDl=0.1, order="reference",
                         return_keys=False):
        pts = np.ascontiguousarray(points, np.float32)
        n = pts.shape[0]
        feats = None if features is None else np.ascontiguousarray(features, np.float32).reshape(n, -1)
        cls = None if classes is None else np.ascontiguousarray(classes, np.int32).reshape(n, -1)
        fdim = 0 if feats is None else feats.shape[1]
        ldim = 0 if cls is None else cls.shape[1]
        op = np.empty((n, 3), np.float32)
        of = np.empty((n, max(fdim, 1)), np.float32)
        oc = np.empty((n, max(ldim, 1)), np.int32)
        ok = np.empty(n, np.uint64)
        m = self.lib.oracle_grid_subsampling(pts, n, _opt(feats), fdim, _opt(cls), ldim, float(sampleDl),
                                             0 if order == "reference" else 1, _opt(op), _opt(of), _opt(oc), _opt(ok))
        out = [op[:m].copy()]
        if feats is not None:
            out.append(of[:m, :fdim].copy())
        if cls is not None:
            out.append(oc[:m, :ldim].copy())
        if return_keys:
            out.append(ok[:m].copy())
        return tuple(out)

    def knn(self, pts, queries, K, return_dist=False):
        p = np.ascontiguousarray(pts, np.float32)
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((q.shape[0], K), np.int64)
        d = np.zeros((q.shape[0], K), np.float32)
        self.lib.oracle_knn(p, p.shape[0], p.shape[1], q, q.shape[0], K, out, _opt(d))
        return (out, d) if return_dist else out

    def knn_batch(self, pts, queries, K, threads=1):
        p = np.ascontiguousarray(pts, np.float32)
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((p.shape[0], q.shape[1], K), np.int64)
        self.lib.oracle_knn_batch(p, p.shape[0], p.shape[1], p.shape[2], q, q.shape[1], K, out, int(threads))
        return out

    def knn_batch_distance_pick(self, pts, nqueries, K, seed):
        p = np.ascontiguousarray(pts, np.float32)
        out = np.zeros((p.shape[0], nqueries, K), np.int64)
        q = np.zeros((p.shape[0], nqueries, p.shape[2]), np.float32)
        self.lib.oracle_knn_batch_distance_pick.restype = None
        self.lib.oracle_knn_batch_distance_pick.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, _i64p, C.c_uint32]
        self.lib.oracle_knn_batch_distance_pick(p, p.shape[0], p.shape[1], p.shape[2], q, nqueries, K, out, int(seed) & 0xffffffff)
        return out, q

    def kdtree(self, pts, leaf_max=10):
        p = np.ascontiguousarray(pts, np.float32)
        n = p.shape[0]
        vind = np.zeros(n, np.int64)
        cap = 2 * n + 8
        ni = np.zeros((cap, 5), np.int32)
        nf = np.zeros((cap, 2), np.float32)
        nn = self.lib.oracle_kdtree_dump(p, n, p.shape[1], leaf_max, vind, _opt(ni), _opt(nf), cap)
        return vind, ni[:nn], nf[:nn]


class _Ref:
    """The reference's own compiled C++ (grid_subsampling.cpp, knn_.cxx) behind oracle/ref_shim.cpp."""

    def __init__(self, path):
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_grid_subsampling.restype = C.c_long
        L.ref_grid_subsampling.argtypes = [_f32p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_float]
        L.ref_grid_subsampling_fetch.restype = None
        L.ref_grid_subsampling_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("ref_knn", "ref_knn_omp"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, _i64p]
        for name in ("ref_knn_batch", "ref_knn_batch_omp"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, _i64p]

    def grid_subsampling(self, points, features=None, classes=None, sampleDl=0.1):
        pts = np.ascontiguousarray(points, np.float32)
        n = pts.shape[0]
        feats = None if features is None else np.ascontiguousarray(features, np.float32).reshape(n, -1)
        cls = None if classes is None else np.ascontiguousarray(classes, np.int32).reshape(n, -1)
        fdim = 0 if feats is None else feats.shape[1]
        ldim = 0 if cls is None else cls.shape[1]
        m = self.lib.ref_grid_subsampling(pts, n, _opt(feats), fdim, _opt(cls), ldim, float(sampleDl))
        op = np.empty((m, 3), np.float32)
        of = np.empty((m, max(fdim, 1)), np.float32)
        oc = np.empty((m, max(ldim, 1)), np.int32)
        self.lib.ref_grid_subsampling_fetch(_opt(op), _opt(of) if fdim else None, _opt(oc) if ldim else None)
        out = [op]
        if feats is not None:
            out.append(of[:, :fdim].copy())
        if cls is not None:
            out.append(oc[:, :ldim].copy())
        return tuple(out)

    def knn(self, pts, queries, K, omp=False):
        p = np.ascontiguousarray(pts, np.float32)
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((q.shape[0], K), np.int64)
        (self.lib.ref_knn_omp if omp else self.lib.ref_knn)(p, p.shape[0], p.shape[1], q, q.shape[0], K, out)
        return out

    def knn_batch(self, pts, queries, K, omp=False):
        p = np.ascontiguousarray(pts, np.float32)
        q = np.ascontiguousarray(queries, np.float32)
        out = np.zeros((p.shape[0], q.shape[1], K), np.int64)
        (self.lib.ref_knn_batch_omp if omp else self.lib.ref_knn_batch)(
            p, p.shape[0], p.shape[1], p.shape[2], q, q.shape[1], K, out)
        return out

    def knn_batch_distance_pick(self, pts, nqueries, K, seed):
        """cpp_knn_batch_distance_pick with the clock it seeds std::mt19937 from pinned to `seed` (oracle/ref_shim.cpp)."""
        L = self.lib
        if not hasattr(L, "ref_knn_batch_distance_pick"):
            return None                                       # an older prebuilt _ref without the hook
        L.ref_set_time.restype = None; L.ref_set_time.argtypes = [C.c_long]
        L.ref_knn_batch_distance_pick.restype = None
        L.ref_knn_batch_distance_pick.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, _i64p]
        p = np.ascontiguousarray(pts, np.float32)
        out = np.zeros((p.shape[0], nqueries, K), np.int64)
        q = np.zeros((p.shape[0], nqueries, p.shape[2]), np.float32)
        L.ref_set_time(int(seed))
        try:
            L.ref_knn_batch_distance_pick(p, p.shape[0], p.shape[1], p.shape[2], q, nqueries, K, out)
        finally:
            L.ref_set_time(-1)
        return out, q


_c = None
_ref = False


def c():
    global _c
    if _c is None:
        _c = _Oracle()
    return _c


def ref():
    """The real reference build, or None when oracle/_ref/libssdr_ref.so does not exist."""
    global _ref
    if _ref is False:
        build()
        path = os.path.join(_HERE, "_ref", "libssdr_ref.so")
        _ref = _Ref(path) if os.path.exists(path) else None
    return _ref
