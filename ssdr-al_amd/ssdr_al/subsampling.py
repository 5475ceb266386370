"""Host-side mirror of the reference's ``grid_subsampling`` CPython module
(/root/reference/SSDR_AL_s3dis/utils/cpp_wrappers/cpp_subsampling/wrapper.cpp:58-276): same keyword
names, coercions, error type/messages and return structure; the work is done by libssdr_al.so."""
import ctypes as C

import numpy as np

from . import _lib


def _coerce(obj, dtype, what):
    try:
        return np.ascontiguousarray(obj, dtype=dtype)
    except Exception:
        # wrapper.cpp:109-132
        raise RuntimeError("Error converting input %s to numpy arrays of type %s"
                           % (what, "int32" if dtype == np.int32 else "float32"))


def compute(points, *, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0, order="reference"):
    """wrapper.cpp:58-80: ``points`` positional, everything else keyword-only.

    Returns ``points`` / ``(points, features)`` / ``(points, classes)`` / ``(points, features, classes)`` with
    shapes [M,3] f32, [M,fdim] f32, [M,ldim] i32 (2-D even for 1-D class input, wrapper.cpp:239-243).
    ``order`` is an extension: "reference" (default, rows in the reference's order) or "key" (ascending voxel
    key, cheaper)."""
    if method not in ("barycenters", "voxelcenters"):          # wrapper.cpp:86-90; otherwise ignored
        raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
    pts = _coerce(points, np.float32, "points")
    feats = None if features is None else _coerce(features, np.float32, "features")
    cls = None if classes is None else _coerce(classes, np.int32, "classes")
    if pts.ndim != 2 or pts.shape[1] != 3:                      # wrapper.cpp:135-142
        raise RuntimeError("Wrong dimensions : points.shape is not (N, 3)")
    if feats is not None and feats.ndim != 2:                   # :143-150
        raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    if cls is not None and cls.ndim > 2:                        # :152-159
        raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
    n = pts.shape[0]
    fdim = feats.shape[1] if feats is not None else 0
    ldim = cls.shape[1] if (cls is not None and cls.ndim == 2) else 1
    if feats is not None and feats.shape[0] != n:               # :175-182
        raise RuntimeError("Wrong dimensions : features.shape is not (N, d)")
    if cls is not None and (cls.ndim == 0 or cls.shape[0] != n):  # :183-190
        raise RuntimeError("Wrong dimensions : classes.shape is not (N,) or (N, d)")
    if n == 0:                                                   # :225-229
        raise RuntimeError("Error")
    L = _lib.lib()
    m = C.c_size_t(0)
    st = L.ssdr_grid_subsample(_lib.ptr(pts), n, _lib.ptr(feats), fdim, _lib.ptr(cls), ldim if cls is not None else 0,
                               float(sampleDl), _lib.ORDER_REFERENCE if order == "reference" else _lib.ORDER_KEY,
                               C.byref(m))
    if st == 4:
        raise RuntimeError("Error")
    _lib.check(st)
    M = m.value
    out_p = np.empty((M, 3), np.float32)
    out_f = np.empty((M, fdim), np.float32) if feats is not None else None
    out_c = np.empty((M, ldim), np.int32) if cls is not None else None
    _lib.check(L.ssdr_grid_subsample_fetch(_lib.ptr(out_p), _lib.ptr(out_f), _lib.ptr(out_c)))
    if feats is not None and cls is not None:                   # wrapper.cpp:269-276
        return out_p, out_f, out_c
    if feats is not None:
        return out_p, out_f
    if cls is not None:
        return out_p, out_c
    return out_p
