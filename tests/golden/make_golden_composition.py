"""Golden vectors for the COMPOSITION of the selection round, produced by running the reference's own
`TSampler.sampling()` (gcn_fps branch) imported from /root/reference (build container only):

    python tests/golden/make_golden_composition.py

What runs unmodified: sampler2.TSampler.sampling -> prediction (:580-642, incl. the population add_clsbal sees),
create_file_top_and_all (:533-552), get_labeled_selection_cloudname_spidx_pointidx (:268-311, GT-dominant member ids,
class-balanced draw), the candidate loop (:745-753), compute_features (:313-342), fps_gcn_cpu.GCN_FPS_sampling.
What is replaced, and only because the image cannot run it: the TensorFlow model (an object whose sess.run hands back
fabricated per-cloud probabilities / features in a shuffled point order, as the real one does through input_inds), the
S3DIS_Dataset / DataLoader pair (an in-order list of cloud names: the reference's own order is a shuffled DataLoader,
sampler2.py:323), and `_help` (label bookkeeping, out of scope).  Inputs live on disk in the reference's own formats
(PLY through helper_ply.write_ply, `.superpoint` / `total.pkl` pickles), fabricated in a temp dir.
Second case: the class-balanced draw alone with more labelled regions than (round_num - 1) * 1000, so the draw is a strict subset.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SSDR_AL_s3dis"


def _stub(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def import_reference():
    np.float = float                      # fps_gcn_cpu.py:64-65 uses the alias NumPy >= 1.24 removed
    for n in ("open3d", "open3d.linux", "torchvision", "torchvision.transforms", "PIL", "PIL.Image", "cpp_wrappers", "cpp_wrappers.cpp_subsampling",
              "cpp_wrappers.cpp_subsampling.grid_subsampling", "nearest_neighbors", "nearest_neighbors.lib",
              "nearest_neighbors.lib.python", "nearest_neighbors.lib.python.nearest_neighbors"):
        _stub(n)
    sys.modules["open3d"].linux = sys.modules["open3d.linux"]
    sys.modules["torchvision.transforms"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "utils"))
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import fps_gcn_cpu
        import sampler2
    finally:
        os.chdir(cwd)
    return sampler2, fps_gcn_cpu


class FakeDataset:
    """stands in for S3DIS_Dataset(mode="sampling"): one item per cloud, in order"""
    names = []

    def __init__(self, **kw):
        self.input_cloud_names = list(FakeDataset.names)

    def __len__(self):
        return len(self.input_cloud_names)


def fake_loader(dataset, batch_size=1, shuffle=False, num_workers=0):
    return list(range(len(dataset)))


class FakeModel:
    """sess.run hands back what the network would: rows in the tile's shuffled order + input_inds to undo it"""

    def __init__(self, probs, feats, perms):
        self.prob_logits, self.last_second_features, self.input_cloud_inds, self.input_input_inds = "P", "F", "C", "I"
        self.sess = self
        self._p, self._f, self._perm = probs, feats, perms

    def get_feed_dict(self, dat, flag):
        return dat

    def run(self, fetch, feed_dict):
        c = int(feed_dict)
        perm = self._perm[c]
        first = self._p[c][perm] if fetch[0] == "P" else self._f[c][perm]
        return first, np.array([c]), perm[None]


def main():
    sampler2, fps_gcn_cpu = import_reference()
    from helper_ply import write_ply
    rng = np.random.default_rng(20261004)
    g = {}
    C = 13
    tmp = tempfile.mkdtemp()
    inp, data = os.path.join(tmp, "input"), os.path.join(tmp, "data")
    os.makedirs(os.path.join(data, "superpoint")); os.makedirs(inp)
    args = ["sb", "WetSU", "clsbal", "gcn_fps"]
    min_size, round_num, batch_size, gcn_number, gcn_top = 4, 3, 24, 1, 0
    names = ["Area_1_office_1", "Area_1_office_2", "Area_2_hall_1"]
    nsps = [60, 45, 30]
    probs, feats, perms, total_obj = [], [], [], {"unlabeled": {}, "selected_class_list": [int(x) for x in rng.integers(0, C, 57)]}
    cur = os.path.join(data, "sampling", sampler2.get_sampler_args_str(args), "round_%d" % (round_num - 1))
    os.makedirs(cur)
    for ci, (name, nsp) in enumerate(zip(names, nsps)):
        szs = rng.integers(2, 40, nsp)                           # some regions fall under min_size
        centres = rng.random((nsp, 3)) * np.array([6.0, 5.0, 2.5])
        xyz = np.concatenate([c + rng.normal(0, 0.15, (s, 3)) for c, s in zip(centres, szs)]).astype(np.float32)
        n = len(xyz)
        perm = rng.permutation(n); inv = np.argsort(perm); xyz = xyz[perm]
        o = np.concatenate([[0], np.cumsum(szs)])
        comps = np.empty(nsp, dtype=object)
        for s in range(nsp):
            comps[s] = list(inv[o[s]:o[s + 1]])
        # ground truth: mostly one class per region, with impurities so that the GT-dominant ids are a strict subset; predictions differ from GT
        gt = np.zeros(n, np.uint8)
        for s in range(nsp):
            base = rng.integers(0, C)
            lab = np.where(rng.random(szs[s]) < 0.7, base, rng.integers(0, C, szs[s]))
            gt[comps[s]] = lab
        write_ply(os.path.join(inp, name + ".ply"), [xyz, np.zeros((n, 3), np.uint8), gt], ["x", "y", "z", "red", "green", "blue", "class"])
        with open(os.path.join(data, "superpoint", name + ".superpoint"), "wb") as f:
            pickle.dump({"components": comps, "in_component": np.zeros(n)}, f)
        p = rng.dirichlet(np.ones(C) * 0.25, n).astype(np.float32)
        probs.append(p); feats.append(rng.normal(0, 1, (n, 32)).astype(np.float32)); perms.append(rng.permutation(n))
        labelled = sorted(rng.choice(nsp, nsp // 4, replace=False).tolist())
        total_obj["unlabeled"][name] = [s for s in range(nsp) if s not in labelled]
        g["a/%d/xyz" % ci], g["a/%d/gt" % ci], g["a/%d/probs" % ci], g["a/%d/feat" % ci] = xyz, gt, p, feats[-1]
        g["a/%d/offsets" % ci] = o.astype(np.int32)
        g["a/%d/points" % ci] = np.concatenate([np.asarray(c, np.int32) for c in comps])
        g["a/%d/labelled" % ci] = np.asarray(labelled, np.int32)
    with open(os.path.join(cur, "total.pkl"), "wb") as f:
        pickle.dump(total_obj, f)
    g["a/selected_class_list"] = np.asarray(total_obj["selected_class_list"], np.int32)
    g["a/params"] = np.asarray([min_size, round_num, batch_size, gcn_number, gcn_top, C], np.int32)

    FakeDataset.names = names
    sampler2.S3DIS_Dataset = FakeDataset
    sampler2.DataLoader = fake_loader
    rec = {}
    sampler2._help = lambda **kw: rec.setdefault("help", []).append((kw["cloud_name"], list(kw["superpoint_inds"])))

    def wrap(mod, fn, key):
        orig = getattr(mod, fn)

        def inner(*a, **kw):
            out = orig(*a, **kw)
            rec[key] = (a, kw, out)
            return out
        setattr(mod, fn, inner)
    wrap(sampler2, "add_clsbal", "clsbal")
    wrap(sampler2, "get_labeled_selection_cloudname_spidx_pointidx", "labsel")
    wrap(sampler2, "compute_features", "features")
    wrap(sampler2, "GCN_FPS_sampling", "gcnfps")
    wrap(fps_gcn_cpu, "farthest_features_sample", "fps")
    orig_pred = sampler2.TSampler.prediction

    def pred(self, **kw):
        out = orig_pred(self, **kw)
        rec["prediction"] = out
        return out
    sampler2.TSampler.prediction = pred

    ts = sampler2.TSampler(input_path=inp, data_path=data, total_num=sum(nsps), test_area_idx=5, sampler_args=args, reg_strength=0.008,
                           min_size=min_size, dataset_name="S3DIS")
    np.random.seed(424242)
    w = {}
    ts.sampling(model=FakeModel(probs, feats, perms), batch_size=batch_size, last_round=round_num - 1, w=w, threshold=0.9, gcn_gpu=0,
                gcn_number=gcn_number, gcn_top=gcn_top)

    ref, sorted_inds, _, lab_dict, class_num = rec["prediction"]
    cid = {n: i for i, n in enumerate(names)}
    g["a/region_cloud"] = np.asarray([cid[r["cloud_name"]] for r in ref], np.int32)          # the population prediction() ranks: unlabelled, >= min_size
    g["a/region_sp"] = np.asarray([r["sp_idx"] for r in ref], np.int32)
    g["a/region_unc"] = np.asarray(rec["clsbal"][2], np.float64)                             # after add_clsbal
    g["a/region_unc_raw"] = np.asarray(rec["clsbal"][1]["region_uncertainty"], np.float64)
    g["a/region_class"] = np.asarray(rec["clsbal"][1]["region_class"], np.int32)
    g["a/sorted_inds"] = np.asarray(sorted_inds, np.int64)
    for ci, n in enumerate(names):
        g["a/%d/labelled_ge_min" % ci] = np.asarray(lab_dict.get(n, []), np.int32)
    labsel, nlab = rec["labsel"][2]
    rows = [(cid[n], sp) for n in labsel for sp in labsel[n]]
    g["a/labsel_cloud"] = np.asarray([r[0] for r in rows], np.int32); g["a/labsel_sp"] = np.asarray([r[1] for r in rows], np.int32)
    ids = [np.asarray(labsel[n][sp], np.int32) for n in labsel for sp in labsel[n]]
    g["a/labsel_ids_off"] = np.concatenate([[0], np.cumsum([len(i) for i in ids])]).astype(np.int32)
    g["a/labsel_ids"] = np.concatenate(ids)                                                   # GT-dominant member ids (cloud point ids)
    lf, lref, uf, uref = rec["features"][2]
    g["a/unl_cloud"] = np.asarray([cid[r["cloud_name"]] for r in uref], np.int32); g["a/unl_sp"] = np.asarray([r["sp_idx"] for r in uref], np.int32)
    g["a/lab_cloud"] = np.asarray([cid[r["cloud_name"]] for r in lref], np.int32); g["a/lab_sp"] = np.asarray([r["sp_idx"] for r in lref], np.int32)
    g["a/unl_feat"] = np.stack(uf).astype(np.float32); g["a/lab_feat"] = np.stack(lf).astype(np.float32)
    g["a/sampling_batch"] = np.int32(rec["gcnfps"][1]["sampling_batch"])
    g["a/fps_seq"] = np.asarray(rec["fps"][2], np.int32)                                     # [0] = the reference's random start
    fl = rec["gcnfps"][2]
    sel = [(cid[n], sp) for n in fl for sp in fl[n]]
    g["a/selected_cloud"] = np.asarray([s[0] for s in sel], np.int32); g["a/selected_sp"] = np.asarray([s[1] for s in sel], np.int32)
    assert sorted(rec["help"]) == sorted((n, list(fl[n])) for n in fl)

    # ---- case b: the class-balanced draw with more labelled regions than the round's budget (round 2: 1000) -------------------------------
    tmp2 = tempfile.mkdtemp()
    inp2, data2 = os.path.join(tmp2, "input"), os.path.join(tmp2, "data")
    os.makedirs(os.path.join(data2, "superpoint")); os.makedirs(inp2)
    lab_dict2 = {}
    for ci, (name, nsp) in enumerate((("cloudX", 700), ("cloudY", 520))):
        szs = rng.integers(1, 6, nsp)
        n = int(szs.sum())
        o = np.concatenate([[0], np.cumsum(szs)])
        perm = rng.permutation(n)
        comps = np.empty(nsp, dtype=object)
        for s in range(nsp):
            comps[s] = list(perm[o[s]:o[s + 1]])
        gt = rng.choice(C, n, p=np.arange(1, C + 1) / np.arange(1, C + 1).sum()).astype(np.uint8)      # skewed classes: the weights matter
        write_ply(os.path.join(inp2, name + ".ply"), [np.zeros((n, 3), np.float32), np.zeros((n, 3), np.uint8), gt], ["x", "y", "z", "red", "green", "blue", "class"])
        with open(os.path.join(data2, "superpoint", name + ".superpoint"), "wb") as f:
            pickle.dump({"components": comps, "in_component": np.zeros(n)}, f)
        lab_dict2[name] = sorted(rng.choice(nsp, nsp - 40, replace=False).tolist())
        g["b/%d/gt" % ci] = gt; g["b/%d/offsets" % ci] = o.astype(np.int32); g["b/%d/points" % ci] = perm.astype(np.int32)
        g["b/%d/labelled" % ci] = np.asarray(lab_dict2[name], np.int32)
    np.random.seed(777)
    sel2, batch2 = sampler2.get_labeled_selection_cloudname_spidx_pointidx(inp2, data2, lab_dict2, C, 2)
    cid2 = {"cloudX": 0, "cloudY": 1}
    rows2 = [(cid2[n], sp) for n in sel2 for sp in sel2[n]]
    g["b/seed"] = np.int32(777); g["b/batch"] = np.int32(batch2)
    g["b/sel_cloud"] = np.asarray([r[0] for r in rows2], np.int32); g["b/sel_sp"] = np.asarray([r[1] for r in rows2], np.int32)
    out = os.path.join(HERE, "composition_golden.npz")
    np.savez_compressed(out, **g)
    print("composition_golden.npz", os.path.getsize(out) // 1024, "KiB; case a: %d ranked regions, %d candidates, %d labelled rows, %d picks; case b: %d of %d drawn"
          % (len(ref), len(uref), len(lref), len(sel), batch2, sum(len(v) for v in lab_dict2.values())))


if __name__ == "__main__":
    main()
