"""The COMPOSITION of the selection round against the reference's own TSampler.sampling() (gcn_fps branch) — golden vectors made by
tests/golden/make_golden_composition.py, which imports sampler2 / fps_gcn_cpu from /root/reference and runs the method on three
fabricated clouds: which regions prediction() ranks (unlabelled, >= min_size), what add_clsbal's histogram is taken over, the
candidate rule, the labelled regions' ground-truth dominant member ids, the class-balanced draw, the features, the picks.

Two layers, as everywhere: the oracle (oracle/pipeline_np.selection_round) against the golden vectors (not gpu), and the product
(ssdr_al.pipeline.HotPath.from_clouds over the C ABI: scoring + the one-call device chain) against the same vectors on the CPU
logic build and on the GPU."""
import numpy as np
import pytest


def _case_a(g):
    min_size, round_num, batch_size, gcn_number, gcn_top, C = [int(x) for x in g["a/params"]]
    clouds, labelled = [], []
    for b in range(3):
        clouds.append(dict(xyz=g["a/%d/xyz" % b], gt=g["a/%d/gt" % b], probs=g["a/%d/probs" % b], feat=g["a/%d/feat" % b],
                           offsets=g["a/%d/offsets" % b], points=g["a/%d/points" % b]))
        labelled.append(set(int(s) for s in g["a/%d/labelled" % b]))
    return clouds, labelled, dict(min_size=min_size, round_num=round_num, batch_size=batch_size, gcn_number=gcn_number, gcn_top=gcn_top, C=C)


def _ref_selected(g):
    return sorted(zip(g["a/selected_cloud"].tolist(), g["a/selected_sp"].tolist()))


def test_oracle_selection_round_against_reference_sampling(golden):
    """oracle/pipeline_np.selection_round == the reference's sampling() run, piece by piece."""
    from oracle import pipeline_np as P
    g = golden("composition_golden.npz")
    clouds, labelled, p = _case_a(g)
    r = P.selection_round(clouds, labelled, g["a/selected_class_list"], p["C"], ["sb", "WetSU", "clsbal", "gcn_fps"], p["min_size"], p["round_num"],
                          p["batch_size"], p["gcn_number"], p["gcn_top"], int(g["a/fps_seq"][0]), np.random.RandomState(424242))
    # the ranked population: unlabelled regions of at least min_size points, cloud by cloud
    assert [b for b, _ in r["region"]] == g["a/region_cloud"].tolist() and [s for _, s in r["region"]] == g["a/region_sp"].tolist()
    assert len(r["region"]) < sum(len(c["offsets"]) - 1 for c in clouds) - sum(len(l) for l in labelled)          # min_size removed some
    assert np.array_equal(r["region_class"], g["a/region_class"])
    assert np.array_equal(r["region_unc_raw"], g["a/region_unc_raw"])                  # WetSU: bit-exact (NumPy's summation order)
    assert np.allclose(r["region_unc"], g["a/region_unc"], rtol=1e-14, atol=0)         # add_clsbal over THIS population + the selected list
    assert np.array_equal(r["sorted_inds"], g["a/sorted_inds"])
    for b in range(3):
        assert r["labelled_ge_min"][b] == g["a/%d/labelled_ge_min" % b].tolist()
    # the labelled draw: same regions in the same draw order, same ground-truth dominant member ids
    assert [(b, s) for b, s, _ in r["labsel"]] != sorted((b, s) for b, s, _ in r["labsel"])      # (a permutation: the draw order is kept)
    got = {(b, s): ids for b, s, ids in r["labsel"]}
    off = g["a/labsel_ids_off"]
    exp = {(int(b), int(s)): g["a/labsel_ids"][off[i]:off[i + 1]] for i, (b, s) in enumerate(zip(g["a/labsel_cloud"], g["a/labsel_sp"]))}
    assert set(got) == set(exp) and all(np.array_equal(got[k], exp[k]) for k in exp)
    # candidates (the reference's order with an in-order loader), features, budget, picks
    assert r["unl"] == list(zip(g["a/unl_cloud"].tolist(), g["a/unl_sp"].tolist()))
    assert np.array_equal(r["unl_feat"].view(np.uint32), g["a/unl_feat"].view(np.uint32))
    lab_ref = {(int(b), int(s)): g["a/lab_feat"][i] for i, (b, s) in enumerate(zip(g["a/lab_cloud"], g["a/lab_sp"]))}
    assert r["lab"] == sorted(lab_ref)
    for i, k in enumerate(r["lab"]):
        assert np.array_equal(r["lab_feat"][i].view(np.uint32), lab_ref[k].view(np.uint32)), k
    assert r["sampling_batch"] == int(g["a/sampling_batch"])
    assert np.array_equal(r["seq"], g["a/fps_seq"])          # same candidate order, same start: the same FPS sequence
    assert sorted(r["selected"]) == _ref_selected(g)


def test_oracle_labelled_draw_is_a_strict_subset(golden):
    """more labelled regions than (round_num - 1) * 1000: the class-balanced draw of sampler2.py:294-302 picks the same 1000"""
    from oracle import pipeline_np as P
    g = golden("composition_golden.npz")
    clouds = [dict(gt=g["b/%d/gt" % b], offsets=g["b/%d/offsets" % b], points=g["b/%d/points" % b]) for b in range(2)]
    lab = [g["b/%d/labelled" % b].tolist() for b in range(2)]
    sel = P.labelled_selection(clouds, lab, 13, 2, np.random.RandomState(int(g["b/seed"])))
    assert len(sel) == int(g["b/batch"]) == 1000 < sum(len(l) for l in lab)
    # the reference groups its draw by cloud (dict of dicts): compare as the same grouping
    mine = [(b, s) for b in range(2) for bb, s, _ in sel if bb == b]
    assert mine == list(zip(g["b/sel_cloud"].tolist(), g["b/sel_sp"].tolist()))


def test_product_labelled_draw(golden, backend):
    """the product's draw (sampler.get_labeled_selection over device-computed ground-truth dominant labels) on the same case"""
    from ssdr_al import sampler
    g = golden("composition_golden.npz")
    doms, refs = [], []
    for b in range(2):
        lab, _ = sampler.dominant_labels(g["b/%d/gt" % b].astype(np.int32), g["b/%d/offsets" % b], g["b/%d/points" % b], 13)
        for s in g["b/%d/labelled" % b]:
            doms.append(int(lab[s])); refs.append((b, int(s)))
    drawn = sampler.get_labeled_selection(doms, 13, 2, np.random.RandomState(int(g["b/seed"])))
    mine = [refs[i] for b in range(2) for i in drawn if refs[i][0] == b]
    assert mine == list(zip(g["b/sel_cloud"].tolist(), g["b/sel_sp"].tolist()))


@pytest.mark.parametrize("rule", ["device", "host"])
def test_product_selection_round_against_reference_sampling(golden, backend, rule, monkeypatch):
    """HotPath.from_clouds: ssdr_point_uncertainty / region_stats / dominant_label / clsbal (masked) / rank + ssdr_gcn_fps_sampling_dev
    (or the host-side rule over the same kernels) select what the reference's sampling() selected."""
    from ssdr_al import pipeline
    from ssdr_al.helper_tool import ConfigS3DIS
    g = golden("composition_golden.npz")
    clouds, labelled, p = _case_a(g)
    if rule == "host":
        monkeypatch.setenv("SSDR_SELECT_HOST_RULE", "1")

    class Cfg(ConfigS3DIS):
        num_classes = p["C"]
    hp = pipeline.HotPath.from_clouds(clouds, labelled, g["a/selected_class_list"], Cfg, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"),
                                      gcn_number=p["gcn_number"], gcn_top=p["gcn_top"], min_size=p["min_size"], round_num=p["round_num"],
                                      label_seed=424242, batch_size=p["batch_size"])
    hp.fps_start = int(g["a/fps_seq"][0])
    sel, unl = hp.step_selection()
    assert hp.rule_path == rule
    base = np.asarray(hp.sp_base)
    # the ranked population and its class-balanced uncertainties
    pop = base[g["a/region_cloud"]] + g["a/region_sp"]
    assert np.array_equal(np.flatnonzero(~hp.skip_mask), pop)
    assert np.allclose(hp.region_unc.to_host()[pop], g["a/region_unc"], rtol=1e-12, atol=0)
    order = hp.sorted_inds.to_host()
    ranked = order[~hp.skip_mask[order]]
    assert np.array_equal(ranked, pop[g["a/sorted_inds"]])
    # labelled rows: the drawn regions (all of the pool here), their ground-truth dominant classes
    rows = sorted((b, s - int(base[b])) for b in hp.lab_rows for s in hp.lab_rows[b])
    assert rows == sorted(zip(g["a/lab_cloud"].tolist(), g["a/lab_sp"].tolist()))
    # candidates in the reference's order, the budget, the picks
    assert [(b, s - int(base[b])) for b, s in unl] == list(zip(g["a/unl_cloud"].tolist(), g["a/unl_sp"].tolist()))
    assert len(sel) == int(g["a/sampling_batch"])
    assert np.array_equal(sel, g["a/fps_seq"])
    assert sorted(hp.selected) == _ref_selected(g)


def test_product_add_classbal_and_masked_clsbal(golden, backend):
    """add_classbal (sampler2.py:256-260) and add_clsbal over a masked population == the reference functions on the population alone"""
    from oracle import select_np as S
    from ssdr_al import sampler
    g = golden("composition_golden.npz")
    rc, raw, sel = g["a/region_class"], g["a/region_unc_raw"], g["a/selected_class_list"]
    assert np.allclose(sampler.add_clsbal(13, rc, raw, {"selected_class_list": list(sel)}), g["a/region_unc"], rtol=1e-13)
    rng = np.random.default_rng(5)
    n = len(rc) + 40
    pos = np.sort(rng.choice(n, len(rc), replace=False))
    full_c = rng.integers(0, 13, n).astype(np.int32); full_u = rng.random(n)
    full_c[pos] = rc; full_u[pos] = raw
    skip = np.ones(n, np.uint8); skip[pos] = 0
    assert np.allclose(sampler.add_clsbal(13, full_c, full_u, {"selected_class_list": list(sel)}, skip)[pos], g["a/region_unc"], rtol=1e-13)
    assert np.allclose(sampler.add_classbal(13, full_c, full_u, skip)[pos], S.add_clsbal(13, rc, raw, ()), rtol=1e-13)


def test_product_labelled_features_use_ground_truth_members(golden, backend):
    """compute_features over the labelled regions (sampler2.py:330-334): the mean over the GT-dominant member ids — ssdr_dominant_label_dev +
    ssdr_segment_mean_features_dev fed the ground-truth pair give the reference's rows bit for bit; the predicted pair does not"""
    from ssdr_al import sampler
    g = golden("composition_golden.npz")
    differs = 0
    for b in range(3):
        gt = g["a/%d/gt" % b].astype(np.int32); off, pts = g["a/%d/offsets" % b], g["a/%d/points" % b]
        rows = [(i, int(s)) for i, (c, s) in enumerate(zip(g["a/lab_cloud"], g["a/lab_sp"])) if c == b]
        sel = np.array([s for _, s in rows], np.int32)
        dom, _ = sampler.dominant_labels(gt, off, pts, 13)
        got = sampler.segment_mean_features(g["a/%d/feat" % b], gt, dom, off, pts, sel)
        exp = g["a/lab_feat"][[i for i, _ in rows]]
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        cls = np.argmax(g["a/%d/probs" % b], -1).astype(np.int32)
        _, pdom, _ = sampler.compute_region_stats(np.zeros(len(cls), np.float32), cls, off, pts, 13, ["WetSU"])
        differs += int((sampler.segment_mean_features(g["a/%d/feat" % b], cls, pdom, off, pts, sel) != exp).any())
    assert differs == 3          # the round-4 composition (predicted members for labelled rows) is a different answer on every cloud


def test_semantic3d_population_filter(golden, backend):
    """the Semantic3D sampler ranks (and draws labelled rows from) regions of at most 1000 points only (SSRD_AL_semantic3d/sampler2.py:644, :655): product ==
    oracle with a size cap on the clouds of the composition case (the cap itself is not pinned by a reference run: the S3D loader feeds parts)"""
    from oracle import pipeline_np as P
    from ssdr_al import pipeline
    from ssdr_al.helper_tool import ConfigS3DIS
    g = golden("composition_golden.npz")
    clouds, labelled, p = _case_a(g)
    cap = 25
    r = P.selection_round(clouds, labelled, g["a/selected_class_list"], p["C"], ["sb", "WetSU", "clsbal", "gcn_fps"], p["min_size"], p["round_num"],
                          p["batch_size"], p["gcn_number"], p["gcn_top"], 0, np.random.RandomState(7), max_size=cap)

    class Cfg(ConfigS3DIS):
        num_classes = p["C"]
    hp = pipeline.HotPath.from_clouds(clouds, labelled, g["a/selected_class_list"], Cfg, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=p["gcn_number"],
                                      gcn_top=p["gcn_top"], min_size=p["min_size"], round_num=p["round_num"], label_seed=7, batch_size=p["batch_size"], max_size=cap)
    sel, unl = hp.step_selection()
    base = np.asarray(hp.sp_base)
    assert 0 < len(r["region"]) < len(g["a/region_sp"])          # the cap removed regions
    assert [(b, s - int(base[b])) for b, s in unl] == r["unl"]
    assert np.array_equal(sel, r["seq"])


def test_semantic3d_chamfer_flavour_in_the_round(golden, backend):
    """the Semantic3D code builds the graph from float32 CUDA-kernel chamfer values (SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30): HotPath(chamfer_mode="f32_cuda")
    through the one-call device chain == the oracle round with the float32 restatement (parity unpinned for the CUDA op), candidates and picks alike"""
    from oracle import pipeline_np as P
    from ssdr_al import pipeline, sampler
    from ssdr_al.helper_tool import ConfigS3DIS
    g = golden("composition_golden.npz")
    clouds, labelled, p = _case_a(g)
    r = P.selection_round(clouds, labelled, g["a/selected_class_list"], p["C"], ["sb", "WetSU", "clsbal", "gcn_fps"], p["min_size"], p["round_num"],
                          p["batch_size"], p["gcn_number"], p["gcn_top"], 0, np.random.RandomState(7), chamfer="f32_cuda")

    class Cfg(ConfigS3DIS):
        num_classes = p["C"]
    hp = pipeline.HotPath.from_clouds(clouds, labelled, g["a/selected_class_list"], Cfg, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=p["gcn_number"],
                                      gcn_top=p["gcn_top"], min_size=p["min_size"], round_num=p["round_num"], label_seed=7, batch_size=p["batch_size"], chamfer_mode="f32_cuda")
    try:
        sel, unl = hp.step_selection()
    finally:
        sampler.set_chamfer_mode("f64")
    base = np.asarray(hp.sp_base)
    assert hp.rule_path == "device"
    assert [(b, s - int(base[b])) for b, s in unl] == r["unl"]
    assert np.array_equal(sel, r["seq"])


def test_round_without_labelled_regions(golden, backend):
    """round 1 of the loop's shape: nothing labelled yet in the clouds at hand (no labelled rows, no draw): product == oracle, FPS and k-center refused
    (kCenterGreedy needs its seeds: the reference's gcn branch is not run before labelled regions exist)"""
    from oracle import pipeline_np as P
    from ssdr_al import pipeline
    from ssdr_al.helper_tool import ConfigS3DIS
    g = golden("composition_golden.npz")
    clouds, labelled, p = _case_a(g)
    none = [set() for _ in clouds]
    r = P.selection_round(clouds, none, g["a/selected_class_list"], p["C"], ["sb", "WetSU", "clsbal", "gcn_fps"], 1, p["round_num"], 30, 1, 0, 0, np.random.RandomState(1))

    class Cfg(ConfigS3DIS):
        num_classes = p["C"]
    hp = pipeline.HotPath.from_clouds(clouds, none, g["a/selected_class_list"], Cfg, sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=1, gcn_top=0,
                                      min_size=1, round_num=p["round_num"], label_seed=1, batch_size=30)
    sel, unl = hp.step_selection()
    base = np.asarray(hp.sp_base)
    assert hp._sel_static["n_lab"] == 0 and len(r["lab"]) == 0
    assert [(b, s - int(base[b])) for b, s in unl] == r["unl"]
    assert np.array_equal(sel, r["seq"])
