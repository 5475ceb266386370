"""`libply_c.compute_geof` and `libply_c.prune` of the reference's partition stage (partition/ply_c/ply_c.cpp:385-455, :289-383) on the GPU."""
import numpy as np

from . import _lib
from ._lib import DevArray


def compute_geof(xyz, target, k_nn):
    """[n,4] float32: linearity, planarity, scattering, verticality of every point with its k_nn neighbours `target`
    (uint32 [n*k_nn], as compute_graph_nn_2 returns them)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    target = np.ascontiguousarray(target, np.uint32).reshape(-1)
    n = xyz.shape[0]
    if target.shape[0] != n * k_nn:
        raise ValueError("target must hold n * k_nn neighbour ids")
    d_x, d_t, d_g = DevArray.from_host(xyz), DevArray.from_host(target), DevArray((n, 4), np.float32)
    _lib.check(_lib.lib().ssdr_geof_dev(d_x.ptr, n, d_t.ptr, k_nn, d_g.ptr, None))
    _lib.sync()
    return d_g.to_host()


def prune(xyz, voxel_size, rgb, labels, objects, n_labels, n_objects):
    """libply_c.prune(xyz f4 [n,3], voxel_size, rgb u1 [n,3], labels u1 [n], objects u4 [n], n_labels, n_objects)
    (ply_c.cpp:289-383, called as in partition/partition.py:126): voxel-grid averages, rows in first-met order.
    Returns (xyz f4 [m,3], rgb u1 [m,3], labels u4 [m, n_labels+1], objects u4 [m, n_objects+1]) — label / object HISTOGRAMS per
    voxel, as the reference returns them (an all-zero single column when n_labels / n_objects is 0)."""
    import ctypes as C
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[0]
    rgb = np.ascontiguousarray(rgb, np.uint8).reshape(n, 3)
    d_x, d_c = DevArray.from_host(xyz), DevArray.from_host(rgb)
    d_l = DevArray.from_host(np.ascontiguousarray(labels, np.uint8).reshape(-1)) if n_labels > 0 else None
    d_o = DevArray.from_host(np.ascontiguousarray(objects, np.uint32).reshape(-1)) if n_objects > 0 else None
    o_x, o_c, o_m = DevArray((n, 3), np.float32), DevArray((n, 3), np.uint8), DevArray((1,), np.int64)
    o_l = DevArray((n, n_labels + 1), np.uint32) if n_labels > 0 else None
    o_o = DevArray((n, n_objects + 1), np.uint32) if n_objects > 0 else None
    _lib.check(_lib.lib().ssdr_prune_dev(d_x.ptr, n, float(voxel_size), d_c.ptr, d_l.ptr if d_l else None, int(n_labels), d_o.ptr if d_o else None, int(n_objects),
                                         o_x.ptr, o_c.ptr, o_l.ptr if o_l else None, o_o.ptr if o_o else None, o_m.ptr, None))
    _lib.check(_lib.lib().ssdr_prune_status(None, None))
    m = int(o_m.to_host()[0])
    lab = o_l.to_host()[:m] if o_l else np.zeros((m, 1), np.uint32)
    obj = o_o.to_host()[:m] if o_o else np.zeros((m, 1), np.uint32)
    return o_x.to_host()[:m], o_c.to_host()[:m], lab, obj
