"""oracle/select_np.py against the golden vectors produced by the reference's own Python (CPU only)."""
import numpy as np

from oracle import select_np as S


def _g(golden):
    return golden("select_golden.npz")


def test_point_uncertainty(golden):
    g = _g(golden)
    for mode in ("lc", "entropy", "sb"):
        assert np.array_equal(S.point_uncertainty(g["u/prob"], mode), g["u/pu_" + mode]), mode


def test_region_stats_and_clsbal(golden):
    g = _g(golden)
    prob = g["u/prob"]; cls = np.argmax(prob, -1); pu = g["u/pu_sb"]
    for mode in ("mean", "sum_weight", "WetSU"):
        ru, dom, cnt = S.region_stats(pu, cls, g["u/offsets"], g["u/points"], 13, mode)
        assert np.allclose(ru, g["u/ru_" + mode], rtol=1e-12, atol=0), mode
        assert np.array_equal(dom, g["u/dom"])
    lab, pur = S.dominant_labels(cls.astype(np.int32), g["u/offsets"], g["u/points"])
    assert np.array_equal(lab, g["u/dom"]) and np.allclose(pur, g["u/purity"])
    cb = S.add_clsbal(13, g["u/dom"], g["u/ru_WetSU"], g["u/selected_class_list"])
    assert np.allclose(cb, g["u/clsbal"], rtol=1e-14)


def _blocks(g):
    out = []
    for name in ("cloudA", "cloudB"):
        xyz, off, pts = g["f/%s/xyz" % name], g["f/%s/offsets" % name], g["f/%s/points" % name]
        cen = S.bbox_centres(xyz, off, pts)
        out.append((xyz, off, pts, cen))
    return out


def test_chamfer_and_adjacency(golden):
    g = _g(golden)
    blocks = _blocks(g)
    for (xyz, off, pts, cen), name in zip(blocks, ("cloudA", "cloudB")):
        cd = S.create_cd(xyz, off, pts, cen)
        assert np.allclose(cd, g["f/%s/cd" % name], rtol=1e-13, atol=1e-15)
    # assemble the global matrix the reference builds: rows = unlabeled refs then labelled refs
    refs = list(zip(g["f/unl_cloud"], g["f/unl_sp"])) + list(zip(g["f/lab_cloud"], g["f/lab_sp"]))
    N = len(refs); adj = np.zeros((N, N))
    for c, (xyz, off, pts, cen) in enumerate(blocks):
        rows = [i for i, (cc, _) in enumerate(refs) if cc == c]
        sp = [refs[i][1] for i in rows]
        sub_off = np.concatenate([[0], np.cumsum([off[s + 1] - off[s] for s in sp])]).astype(np.int32)
        sub_pts = np.concatenate([pts[off[s]:off[s + 1]] for s in sp])
        cen_s = S.bbox_centres(xyz, sub_off, sub_pts)
        A = S.block_adjacency(cen_s, S.create_cd(xyz, sub_off, sub_pts, cen_s))
        adj[np.ix_(rows, rows)] = A
    assert np.allclose(adj, g["f/adj"], rtol=1e-12, atol=1e-15)
    V = np.concatenate([g["f/unl_feat"], g["f/lab_feat"]]).astype(np.float64)
    for gn in (1, 2, 3):
        comb = S.propagate([adj], [np.arange(N)], V, gn)
        seq = S.farthest_features_sample(comb[:9], 5, int(g["f/gcnfps_start_%d" % gn]))
        a = [int(g["f/unl_sp"][i]) for i in seq if g["f/unl_cloud"][i] == 0]
        b = [int(g["f/unl_sp"][i]) for i in seq if g["f/unl_cloud"][i] == 1]
        assert a == list(g["f/gcnfps_A_%d" % gn]) and b == list(g["f/gcnfps_B_%d" % gn])
    # keep-top mask (fps_gcn_cpu.py:153-160), block by block against the masked global matrix of the reference
    for gt in (2, 3, 5, 13):
        masked = np.zeros((N, N))
        for c in range(2):
            rows = [i for i, (cc, _) in enumerate(refs) if cc == c]
            masked[np.ix_(rows, rows)] = S.keep_top(adj[np.ix_(rows, rows)], gt)
        assert np.allclose(masked, g["f/adj_top%d" % gt], rtol=1e-12, atol=1e-15)
        for gn in (1, 2):
            seq = S.farthest_features_sample(S.propagate([masked], [np.arange(N)], V, gn)[:9], 5, int(g["f/gcnfps_start_%d" % gn]))
            a = [int(g["f/unl_sp"][i]) for i in seq if g["f/unl_cloud"][i] == 0]
            b = [int(g["f/unl_sp"][i]) for i in seq if g["f/unl_cloud"][i] == 1]
            assert a == list(g["f/gcnfps_top%d_A_%d" % (gt, gn)]) and b == list(g["f/gcnfps_top%d_B_%d" % (gt, gn)])


def test_gcn_top_100(golden):
    """the oracle against the reference's run with its scripts' own --gcn_top 100 (70 + 60 superpoints)"""
    g = _g(golden)
    names = ("cloudC", "cloudD")
    refs = list(zip(g["g/unl_cloud"], g["g/unl_sp"])) + list(zip(g["g/lab_cloud"], g["g/lab_sp"]))
    N, nu = len(refs), len(g["g/unl_sp"])
    adj = np.zeros((N, N))
    for c, name in enumerate(names):
        xyz, off, pts = g["g/%s/xyz" % name], g["g/%s/offsets" % name], g["g/%s/points" % name]
        rows = [i for i, (cc, _) in enumerate(refs) if cc == c]
        sp = [refs[i][1] for i in rows]
        sub_off = np.concatenate([[0], np.cumsum([off[s + 1] - off[s] for s in sp])]).astype(np.int32)
        sub_pts = np.concatenate([pts[off[s]:off[s + 1]] for s in sp])
        cen_s = S.bbox_centres(xyz, sub_off, sub_pts)
        adj[np.ix_(rows, rows)] = S.block_adjacency(cen_s, S.create_cd(xyz, sub_off, sub_pts, cen_s))
    V = np.concatenate([g["g/unl_feat"], g["g/lab_feat"]]).astype(np.float64)
    for gt in (5, 100):
        masked = np.zeros((N, N))
        for c in range(2):
            rows = [i for i, (cc, _) in enumerate(refs) if cc == c]
            masked[np.ix_(rows, rows)] = S.keep_top(adj[np.ix_(rows, rows)], gt)
        assert np.allclose(masked, g["g/adj_top%d" % gt], rtol=2e-7, atol=1e-12)
        for gn in (1, 2):
            seq = S.farthest_features_sample(S.propagate([masked], [np.arange(N)], V, gn)[:nu], 20, int(g["g/start"]))
            a = [int(g["g/unl_sp"][i]) for i in seq if g["g/unl_cloud"][i] == 0]
            b = [int(g["g/unl_sp"][i]) for i in seq if g["g/unl_cloud"][i] == 1]
            assert a == list(g["g/gcnfps_top%d_C_%d" % (gt, gn)]) and b == list(g["g/gcnfps_top%d_D_%d" % (gt, gn)])


def test_fps_and_kcenter_sequences(golden):
    g = _g(golden)
    assert np.array_equal(S.farthest_features_sample(g["fps/feat"], 50, int(g["fps/start"])), g["fps/seq"])
    assert np.array_equal(S.kcenter_greedy(g["kc/feat"], g["kc/already"], 30), g["kc/seq"])


def test_oracle_create_adj_matches_reference_golden(golden):
    """oracle/select_np.py:create_adj against gcn.create_adj's own output (tests/golden/make_golden_gcn.py)."""
    from oracle import select_np as S
    g = golden("select_golden.npz"); G = golden("gcn_golden.npz")
    names = ["cloudC", "cloudD"]
    refs = [(int(c), int(s)) for c, s in zip(g["g/unl_cloud"], g["g/unl_sp"])] + [(int(c), int(s)) for c, s in zip(g["g/lab_cloud"], g["g/lab_sp"])]
    cens, cds, rows = [], [], []
    for ci, n in enumerate(names):
        xyz, off, pts = g["g/%s/xyz" % n], g["g/%s/offsets" % n], g["g/%s/points" % n]
        sel = [s for c, s in refs if c == ci]
        o2 = np.concatenate([[0], np.cumsum([off[s + 1] - off[s] for s in sel])]).astype(np.int64)
        p2 = np.concatenate([pts[off[s]:off[s + 1]] for s in sel])
        cen = S.bbox_centres(xyz, o2, p2)
        cens.append(cen); cds.append(S.create_cd(xyz, o2, p2, cen)); rows.append([i for i, (c, _) in enumerate(refs) if c == ci])
    V, adj = S.create_adj(np.concatenate([g["g/unl_feat"], g["g/lab_feat"]]), cens, cds, rows)
    assert np.abs(V - G["featuresV"]).max() < 1e-6
    assert (np.abs(adj - G["adj"]) / np.maximum(np.abs(G["adj"]), 1.0)).max() < 2e-4      # float32 column sums near 0 (see tests/test_select.py)
