"""Golden files of the on-disk formats, written by the REFERENCE's own code (helper_ply.write_ply, the pickle layout of
partition/compute_superpoint.py:63-87).  Run in the build container: python tests/golden/make_golden_formats.py"""
import os, pickle, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/SSDR_AL_s3dis")
import helper_ply as ref          # pure Python + NumPy: importable here

rng = np.random.default_rng(7)
n = 57
xyz = rng.normal(0, 2, (n, 3)).astype(np.float32)
rgb = rng.integers(0, 256, (n, 3)).astype(np.uint8)
lab = rng.integers(0, 13, n).astype(np.uint8)
out = os.path.join(HERE, "formats")
os.makedirs(out, exist_ok=True)
ref.write_ply(os.path.join(out, "cloud.ply"), [xyz, rgb, lab], ["x", "y", "z", "red", "green", "blue", "class"])
faces = rng.integers(0, n, (11, 3)).astype(np.int32)
ref.write_ply(os.path.join(out, "mesh.ply"), [xyz.astype(np.float64), lab.astype(np.int32)], ["x", "y", "z", "scalar"], triangular_faces=faces)
# the partition stage's pickles (compute_superpoint.py:63-74): components as libcp returns them (list of index lists)
components = [list(map(int, np.flatnonzero(np.arange(n) % 5 == k))) for k in range(5)]
in_component = (np.arange(n) % 5).astype(np.uint32)
comp = np.array(components, dtype="object")
with open(os.path.join(out, "cloud.superpoint"), "wb") as f:
    pickle.dump({"components": comp, "in_component": in_component}, f)
with open(os.path.join(out, "cloud.gt"), "wb") as f:
    pickle.dump(np.zeros([2, n], dtype=np.float32), f)
np.savez(os.path.join(out, "expected.npz"), xyz=xyz, rgb=rgb, lab=lab, faces=faces, in_component=in_component,
         comp_flat=np.concatenate(components), comp_len=np.array([len(c) for c in components]))
print("wrote", sorted(os.listdir(out)))
