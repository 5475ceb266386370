"""`libply_c.compute_geof` of the reference's partition stage (partition/ply_c/ply_c.cpp:385-455) on the GPU."""
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
