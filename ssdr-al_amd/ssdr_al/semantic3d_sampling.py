"""Host-side mirror of the part the Semantic3D sampling loader plays in front of the path
(/root/reference/SSRD_AL_semantic3d/semantic3d_dataset_sampling.py): `split3` (:198-236) and the merge rule of `tf_map` (:243-255) that cut
a whole scan into the parts the KNN pyramid and the network are run on (at most 800 000 points each).  The partition runs in
libssdr_al.so (ssdr_split3_dev); this module moves the arrays."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DevArray


def split3_parts(batch_xyz, max_size=800000, merge_max=2000, return_part_ids=False, recurse_max_size=800000):
    """batch_xyz [n,3] -> the combined parts tf_map loops over (:253), each an int32 array of point indices: ascending inside a leaf, leaves in
    the order split3 appends them (the reference's own order inside a part is CPython's set iteration order).  `max_size` is the top-level
    call's; the parts of a split part are held against `recurse_max_size` (the reference's recursive call passes the literal 800000, :233).  Raises for a part above max_size
    whose points are coincident (the reference recurses without end)."""
    xyz = np.ascontiguousarray(batch_xyz, np.float32)
    n = len(xyz)
    d_x = DevArray.from_host(xyz); d_part = DevArray((n,), np.int32); d_order = DevArray((n,), np.int32)
    cap = 4096
    off = np.zeros(cap + 1, np.int64); num = C.c_size_t()
    _lib.check(_lib.lib().ssdr_split3_dev(d_x.ptr, n, int(max_size), int(recurse_max_size), int(merge_max), d_part.ptr, d_order.ptr, _lib.ptr(off), cap, C.byref(num), None))
    order = d_order.to_host()
    parts = [order[off[k]:off[k + 1]] for k in range(num.value)]
    return (parts, d_part.to_host()) if return_part_ids else parts
