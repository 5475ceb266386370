"""Golden vectors for the trained-GCN branch's adjacency, produced by the reference's OWN gcn.create_adj (gcn.py:116-191) imported from
/root/reference (build container only):  python tests/golden/make_golden_gcn.py

create_adj reads its clouds from disk (.superpoint pickles + PLYs): the 70 + 60 superpoint fixture of select_golden.npz ("g/...") is written
to a temp dir with the reference's own helper_ply.write_ply.  The function moves its tensors to a CUDA device; there is none here, so the
generator makes `.cuda(...)` the identity on torch tensors (the reference file itself is untouched): the arithmetic is torch float32 on the CPU."""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SSDR_AL_s3dis"


def main():
    import torch
    np.float = float                                   # gcn.py:141 uses the alias NumPy >= 1.24 removed
    torch.Tensor.cuda = lambda self, *a, **k: self     # no CUDA device in the build container
    sys.path.insert(0, REF); sys.path.insert(0, os.path.join(REF, "utils"))
    cwd = os.getcwd(); os.chdir(REF)
    try:
        import gcn
    finally:
        os.chdir(cwd)
    from helper_ply import write_ply
    G = np.load(os.path.join(HERE, "select_golden.npz"))
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "superpoint")); os.makedirs(os.path.join(tmp, "input"))
    names = ["cloudC", "cloudD"]
    for name in names:
        xyz, o, pts = G["g/%s/xyz" % name], G["g/%s/offsets" % name], G["g/%s/points" % name]
        comps = np.empty(len(o) - 1, dtype=object)
        for s in range(len(o) - 1):
            comps[s] = list(pts[o[s]:o[s + 1]])
        write_ply(os.path.join(tmp, "input", name + ".ply"), [xyz, np.zeros((len(xyz), 3), np.uint8), np.zeros(len(xyz), np.uint8)],
                  ["x", "y", "z", "red", "green", "blue", "class"])
        with open(os.path.join(tmp, "data", "superpoint", name + ".superpoint"), "wb") as f:
            pickle.dump({"components": comps, "in_component": np.zeros(len(xyz))}, f)
    unl = [{"cloud_name": names[c], "sp_idx": int(s)} for c, s in zip(G["g/unl_cloud"], G["g/unl_sp"])]
    lab = [{"cloud_name": names[c], "sp_idx": int(s)} for c, s in zip(G["g/lab_cloud"], G["g/lab_sp"])]
    featuresV = np.concatenate([G["g/unl_feat"], G["g/lab_feat"]])          # gcn.py:199
    V, adj, _ = gcn.create_adj(featuresV, lab, unl, os.path.join(tmp, "input"), os.path.join(tmp, "data"), 0)
    out = {"featuresV": V.numpy(), "adj": adj.numpy()}
    np.savez_compressed(os.path.join(HERE, "gcn_golden.npz"), **out)
    print("gcn_golden.npz", os.path.getsize(os.path.join(HERE, "gcn_golden.npz")) // 1024, "KiB", out["adj"].shape, out["adj"].dtype)


if __name__ == "__main__":
    main()
