"""Golden vectors of the superpoint-graph k-NN structures, produced by the REFERENCE's own partition/graphs.py
(importable here: NumPy + scikit-learn + SciPy).  Run in the build container: python tests/golden/make_golden_graph.py"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/SSDR_AL_s3dis/partition")
import graphs as ref

rng = np.random.default_rng(11)
out = {}
for name, n in (("uniform", 500), ("room", 900)):
    x = (rng.random((n, 3)) * np.array([5, 4, 3])).astype(np.float32)
    if name == "room":                      # two planes + clutter, like a subsampled room
        x[: n // 2, 2] = rng.normal(0, 0.003, n // 2).astype(np.float32)
        x[n // 2: 3 * n // 4, 0] = rng.normal(0, 0.003, 3 * n // 4 - n // 2).astype(np.float32)
    g, t2 = ref.compute_graph_nn_2(x, 10, 45)
    out[name + "/xyz"] = x
    out[name + "/source"], out[name + "/target"], out[name + "/distances"], out[name + "/target2"] = g["source"], g["target"], g["distances"], t2
    g1 = ref.compute_graph_nn(x, 7)
    out[name + "/nn7_target"], out[name + "/nn7_distances"] = g1["target"], g1["distances"]
# the voronoi > 0 branch (:38-62) reads `tri.vertices`, which SciPy 1.11 renamed to `simplices`: hand the reference a Delaunay with the old name
from scipy.spatial import Delaunay as _Delaunay


class _OldDelaunay(_Delaunay):
    @property
    def vertices(self):
        return self.simplices


ref.Delaunay = _OldDelaunay
x = out["uniform/xyz"][:300]
g, t2 = ref.compute_graph_nn_2(x, 5, 12, voronoi=0.3)
out["voronoi/source"], out["voronoi/target"], out["voronoi/distances"], out["voronoi/target2"] = g["source"], g["target"], g["distances"], t2
np.savez_compressed(os.path.join(HERE, "graph_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
