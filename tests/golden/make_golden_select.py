"""Golden vectors for the selection stage, produced by the reference's OWN Python functions imported from
/root/reference (build container only):  python tests/golden/make_golden_select.py

Imported: sampler2.{compute_point_uncertainty, compute_region_uncertainty, _dominant_label, add_clsbal},
fps_gcn_cpu.{create_cd, fps_adj_all, GCN_FPS_sampling, farthest_features_sample}, kcenterGreedy.kCenterGreedy.
Modules the image lacks (open3d, torchvision) and the reference's compiled ops are replaced by empty stand-in
modules ONLY so that `import sampler2` succeeds; none of the functions exercised touches them.
fps_adj_all reads pickles/PLYs from disk, so small fixture files are fabricated in a temp dir with the
reference's own helper_ply.write_ply.
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


def main():
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
    import fps_gcn_cpu
    import kcenterGreedy
    try:
        import sampler2
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(77)
    g = {}

    # U: probabilities + ragged superpoints
    n, C = 3000, 13
    prob = rng.dirichlet(np.ones(C) * 0.3, n).astype(np.float32)
    prob[:5] = 0; prob[:5, 2] = 1                       # exact zeros -> log2 = -inf branch
    g["u/prob"] = prob
    for mode in ("lc", "entropy", "sb"):
        g["u/pu_" + mode] = sampler2.compute_point_uncertainty(prob, [mode])
    cls = np.argmax(prob, -1)
    sizes = rng.integers(3, 60, 80); sizes[-1] = n - sizes[:-1].sum()
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pts = rng.permutation(n).astype(np.int32)
    g["u/offsets"], g["u/points"] = offs, pts
    pu = g["u/pu_sb"]
    for mode in ("mean", "sum_weight", "WetSU"):
        g["u/ru_" + mode] = np.array([sampler2.compute_region_uncertainty(pu[pts[offs[s]:offs[s + 1]]], cls[pts[offs[s]:offs[s + 1]]], C, [mode])
                                      for s in range(80)], np.float64)
    dom = [sampler2._dominant_label(cls[pts[offs[s]:offs[s + 1]]]) for s in range(80)]
    g["u/dom"] = np.array([d[0] for d in dom], np.int32)
    g["u/purity"] = np.array([d[1] for d in dom], np.float64)
    # compute_features' per-superpoint mean (sampler2.py:333, :339): np.mean(last_second_features[dominant_point_ids], axis=0) with the
    # dominant_point_ids the reference derives by _dominant_2 over the predicted classes (:625-626, ids are positions in the list)
    feat = rng.normal(0, 1, (n, 32)).astype(np.float32)
    g["u/feat32"] = feat
    means = []
    for s in range(80):
        ids = pts[offs[s]:offs[s + 1]]
        _, local = sampler2._dominant_2(cls[ids])
        means.append(np.mean(feat[ids[local[0]]], axis=0))
    g["u/segmean"] = np.stack(means).astype(np.float32)
    sel = rng.integers(0, C, 40)
    g["u/selected_class_list"] = sel
    g["u/clsbal"] = sampler2.add_clsbal(C, g["u/dom"], g["u/ru_WetSU"], {"selected_class_list": list(sel)})

    # F: two clouds, 7 + 6 superpoints, fabricated on disk exactly as the reference expects
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "data", "superpoint")); os.makedirs(os.path.join(tmp, "input"))
    from helper_ply import write_ply
    clouds = {}
    for ci, (name, nsp) in enumerate((("cloudA", 7), ("cloudB", 6))):
        szs = rng.integers(5, 40, nsp)
        centres = rng.random((nsp, 3)) * 3
        xyz = np.concatenate([c + rng.normal(0, 0.15, (s, 3)) for c, s in zip(centres, szs)]).astype(np.float32)
        perm = rng.permutation(len(xyz)); inv = np.argsort(perm); xyz = xyz[perm]
        o = np.concatenate([[0], np.cumsum(szs)])
        comps = np.empty(nsp, dtype=object)
        for s in range(nsp):
            comps[s] = list(inv[o[s]:o[s + 1]])
        write_ply(os.path.join(tmp, "input", name + ".ply"), [xyz, np.zeros((len(xyz), 3), np.uint8), np.zeros(len(xyz), np.uint8)],
                  ["x", "y", "z", "red", "green", "blue", "class"])
        with open(os.path.join(tmp, "data", "superpoint", name + ".superpoint"), "wb") as f:
            pickle.dump({"components": comps, "in_component": np.zeros(len(xyz))}, f)
        clouds[name] = (xyz, comps)
        g["f/%s/xyz" % name] = xyz
        g["f/%s/offsets" % name] = o.astype(np.int32)
        g["f/%s/points" % name] = np.concatenate([np.asarray(c, np.int32) for c in comps])
        cen = np.stack([(xyz[c].min(0) + xyz[c].max(0)).astype(np.float64) / 2.0 for c in comps])   # as fps_gcn_cpu.py:86-88 evaluates it
        g["f/%s/cd" % name] = fps_gcn_cpu.create_cd([xyz[c] for c in comps], cen)
    # unlabeled candidates: A0..A4, B0..B3; labelled: A5, A6, B4, B5 (order as the sampler builds it)
    unl = [{"cloud_name": "cloudA", "sp_idx": i} for i in range(5)] + [{"cloud_name": "cloudB", "sp_idx": i} for i in range(4)]
    lab = [{"cloud_name": "cloudA", "sp_idx": 5}, {"cloud_name": "cloudB", "sp_idx": 4}, {"cloud_name": "cloudA", "sp_idx": 6}, {"cloud_name": "cloudB", "sp_idx": 5}]
    adj, _ = fps_gcn_cpu.fps_adj_all(lab, unl, os.path.join(tmp, "input"), os.path.join(tmp, "data"))
    g["f/adj"] = adj
    g["f/unl_cloud"] = np.array([0] * 5 + [1] * 4, np.int32); g["f/unl_sp"] = np.array(list(range(5)) + list(range(4)), np.int32)
    g["f/lab_cloud"] = np.array([0, 1, 0, 1], np.int32); g["f/lab_sp"] = np.array([5, 4, 6, 5], np.int32)
    uf = rng.normal(0, 1, (9, 32)).astype(np.float32); lf = rng.normal(0, 1, (4, 32)).astype(np.float32)
    g["f/unl_feat"], g["f/lab_feat"] = uf, lf
    for gn in (1, 2, 3):
        np.random.seed(5)
        start = np.random.randint(0, 9)       # the draw farthest_features_sample makes first (:133)
        np.random.seed(5)
        fl = fps_gcn_cpu.GCN_FPS_sampling(list(lf), lab, list(uf), unl, os.path.join(tmp, "input"), os.path.join(tmp, "data"), 5, gn, 0)
        g["f/gcnfps_start_%d" % gn] = np.int32(start)
        g["f/gcnfps_A_%d" % gn] = np.array(fl.get("cloudA", []), np.int32)
        g["f/gcnfps_B_%d" % gn] = np.array(fl.get("cloudB", []), np.int32)
    # gcn_top > 0 (the reference's scripts run --gcn_top 100, S3/run_graph_reasoning_analysis.sh:9-11): keep-top mask of :153-160.
    # 2 and 3 cut inside the 7- and 6-row cloud blocks, 5 cuts them less, 13 (= N) keeps everything; gcn_top > N raises in the reference (:159).
    for gt in (2, 3, 5, 13):
        # the masked adjacency GCN_FPS_sampling builds internally (:155-160, the same NumPy statements applied to the reference's adj)
        mask = np.zeros(adj.shape)
        mask[np.repeat(np.expand_dims(np.arange(adj.shape[0]), axis=1), repeats=gt, axis=1), np.argsort(adj, axis=1)[:, -gt:]] = 1.0
        g["f/adj_top%d" % gt] = np.multiply(adj, mask)
        for gn in (1, 2):
            np.random.seed(5)
            fl = fps_gcn_cpu.GCN_FPS_sampling(list(lf), lab, list(uf), unl, os.path.join(tmp, "input"), os.path.join(tmp, "data"), 5, gn, gt)
            g["f/gcnfps_top%d_A_%d" % (gt, gn)] = np.array(fl.get("cloudA", []), np.int32)
            g["f/gcnfps_top%d_B_%d" % (gt, gn)] = np.array(fl.get("cloudB", []), np.int32)

    # FPS / k-center sequences on generic features
    feats = rng.normal(0, 1, (300, 32)).astype(np.float32).astype(np.float64)
    np.random.seed(11); start = np.random.randint(0, 300); np.random.seed(11)
    g["fps/feat"], g["fps/start"] = feats.astype(np.float32), np.int32(start)
    g["fps/seq"] = fps_gcn_cpu.farthest_features_sample(list(feats), 50)
    kf = rng.normal(0, 1, (160, 129)).astype(np.float32).astype(np.float64)
    already = np.arange(120, 160)
    kc = kcenterGreedy.kCenterGreedy(kf)
    g["kc/feat"], g["kc/already"] = kf.astype(np.float32), already.astype(np.int32)
    g["kc/seq"] = np.array(kc.select_batch_(already, 30), np.int32)
    # F6: the "edcd" branch's farthest_superpoint_sample (sampler2.py:49-80) on cloudA's superpoints
    xyzA, compsA = clouds["cloudA"]
    cenA = np.stack([(xyzA[c].min(0) + xyzA[c].max(0)).astype(np.float64) / 2.0 for c in compsA])
    kf2 = rng.normal(0, 1, (1000, 129)).astype(np.float32).astype(np.float64)
    kc2 = kcenterGreedy.kCenterGreedy(kf2)
    g["kc2/feat"] = kf2.astype(np.float32)
    g["kc2/already"] = np.arange(900, 1000, dtype=np.int64)
    g["kc2/seq"] = np.array(kc2.select_batch_(g["kc2/already"], 80), np.int32)
    g["f6/seq"] = sampler2.farthest_superpoint_sample([xyzA[c] for c in compsA], cenA, 5, 0)
    # G: 70 + 60 superpoints, so that the reference scripts' own --gcn_top 100 (S3/run_graph_reasoning_analysis.sh:9-11) is a valid mask
    # width (the reference's mask assignment needs gcn_top <= #regions, :159) and cuts inside a 130-row matrix
    tmp2 = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp2, "data", "superpoint")); os.makedirs(os.path.join(tmp2, "input"))
    big = {}
    for name, nsp in (("cloudC", 70), ("cloudD", 60)):
        szs = rng.integers(4, 30, nsp)
        centres = rng.random((nsp, 3)) * np.array([6.0, 5.0, 2.5])
        xyz = np.concatenate([c + rng.normal(0, 0.12, (s, 3)) for c, s in zip(centres, szs)]).astype(np.float32)
        perm = rng.permutation(len(xyz)); inv = np.argsort(perm); xyz = xyz[perm]
        o = np.concatenate([[0], np.cumsum(szs)])
        comps = np.empty(nsp, dtype=object)
        for s in range(nsp):
            comps[s] = list(inv[o[s]:o[s + 1]])
        write_ply(os.path.join(tmp2, "input", name + ".ply"), [xyz, np.zeros((len(xyz), 3), np.uint8), np.zeros(len(xyz), np.uint8)],
                  ["x", "y", "z", "red", "green", "blue", "class"])
        with open(os.path.join(tmp2, "data", "superpoint", name + ".superpoint"), "wb") as f:
            pickle.dump({"components": comps, "in_component": np.zeros(len(xyz))}, f)
        big[name] = (xyz, comps)
        g["g/%s/xyz" % name] = xyz
        g["g/%s/offsets" % name] = o.astype(np.int32)
        g["g/%s/points" % name] = np.concatenate([np.asarray(c, np.int32) for c in comps])
    # candidates: C0..C54, D0..D44; labelled: C55..C69 and D45..D59 interleaved (the order the sampler builds is arbitrary)
    unl2 = [{"cloud_name": "cloudC", "sp_idx": i} for i in range(55)] + [{"cloud_name": "cloudD", "sp_idx": i} for i in range(45)]
    lab2 = []
    for i in range(15):
        lab2 += [{"cloud_name": "cloudC", "sp_idx": 55 + i}, {"cloud_name": "cloudD", "sp_idx": 45 + i}]
    adj2, _ = fps_gcn_cpu.fps_adj_all(lab2, unl2, os.path.join(tmp2, "input"), os.path.join(tmp2, "data"))
    names2 = ["cloudC", "cloudD"]
    g["g/unl_cloud"] = np.array([names2.index(r["cloud_name"]) for r in unl2], np.int32); g["g/unl_sp"] = np.array([r["sp_idx"] for r in unl2], np.int32)
    g["g/lab_cloud"] = np.array([names2.index(r["cloud_name"]) for r in lab2], np.int32); g["g/lab_sp"] = np.array([r["sp_idx"] for r in lab2], np.int32)
    uf2 = rng.normal(0, 1, (len(unl2), 32)).astype(np.float32); lf2 = rng.normal(0, 1, (len(lab2), 32)).astype(np.float32)
    g["g/unl_feat"], g["g/lab_feat"] = uf2, lf2
    for gt in (5, 100):
        mask = np.zeros(adj2.shape)
        mask[np.repeat(np.expand_dims(np.arange(adj2.shape[0]), axis=1), repeats=gt, axis=1), np.argsort(adj2, axis=1)[:, -gt:]] = 1.0
        g["g/adj_top%d" % gt] = np.multiply(adj2, mask).astype(np.float32)      # compared at 1e-6: float32 keeps the fixture small
        for gn in (1, 2):
            np.random.seed(9)
            start = np.random.randint(0, len(unl2))
            np.random.seed(9)
            fl = fps_gcn_cpu.GCN_FPS_sampling(list(lf2), lab2, list(uf2), unl2, os.path.join(tmp2, "input"), os.path.join(tmp2, "data"), 20, gn, gt)
            g["g/start"] = np.int32(start)
            g["g/gcnfps_top%d_C_%d" % (gt, gn)] = np.array(fl.get("cloudC", []), np.int32)
            g["g/gcnfps_top%d_D_%d" % (gt, gn)] = np.array(fl.get("cloudD", []), np.int32)
    np.savez_compressed(os.path.join(HERE, "select_golden.npz"), **g)
    print("select_golden.npz", os.path.getsize(os.path.join(HERE, "select_golden.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
