"""End-to-end hot path (ssdr_al/pipeline.py over the C ABI) against the oracle pipeline, stage by stage."""
import numpy as np
import pytest

from conftest import assert_bits_equal


def _setup(num_points, nrooms, density, select_per_tile, labeled_per_tile):
    from oracle import randla_np as R
    from ssdr_al import pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS

    class Cfg(ConfigS3DIS):
        pass
    Cfg.num_points = num_points
    W = R.init_weights(0)
    rooms = [synthetic.make_room(5000 + i, density=density) for i in range(nrooms)]
    hp = pipeline.HotPath(W, Cfg, select_per_tile=select_per_tile, labeled_per_tile=labeled_per_tile).load_rooms(rooms)
    return hp, rooms, W


def _ranking_matches(hp, ref):
    """the product ranks every region and masks the ones outside prediction()'s population; the oracle (like the reference) ranks the population"""
    pop = np.array([s for _, s in ref["region"]], np.int64)
    order = hp.sorted_inds.to_host()
    return (np.array_equal(np.flatnonzero(~hp.skip_mask), pop) and np.allclose(hp.region_unc.to_host()[pop], ref["region_unc"], rtol=1e-12, atol=0)
            and np.array_equal(order[~hp.skip_mask[order]], ref["ranked"]))


def test_hot_path_matches_oracle_stage_by_stage(backend):
    from oracle import pipeline_np
    if backend == "emu":
        hp, rooms, W = _setup(2048, 2, 150.0, 6, 3)
    else:
        hp, rooms, W = _setup(40960, 3, 2500.0, 37, 15)
    sel, unl = hp.step()
    ref = pipeline_np.run(hp, rooms, W, threads=4)
    # geometry + indices: bit-exact
    assert_bits_equal(hp.xyz.to_host(), ref["xyz"], "tile xyz")
    assert_bits_equal(hp.feat.to_host(), ref["feat"], "tile features")
    for i in range(5):
        assert_bits_equal(hp.neigh[i].to_host(), ref["neigh"][i], "neigh%d" % i)
        assert_bits_equal(hp.interp[i].to_host(), ref["interp"][i], "interp%d" % i)
    # network: 1e-3 absolute (north_star tolerance, fp32)
    gp, gf = hp.probs.to_host(), hp.f32.to_host()
    assert np.abs(gp - ref["probs"]).max() < 1e-3 and np.abs(gf - ref["f32"]).max() < 1e-3
    # selection: exact given the same network outputs
    ref2 = pipeline_np.run(hp, rooms, W, threads=4, net_outputs=(gp, gf))
    assert_bits_equal(hp.unc.to_host(), ref2["unc"], "point uncertainty")
    assert np.array_equal(hp.cls.to_host(), ref2["cls"])
    assert np.array_equal(hp.tile_l.to_host().reshape(ref2["labels"].shape), ref2["labels"])      # queried_pc_label
    assert _ranking_matches(hp, ref2)
    assert unl == ref2["unl"]
    assert np.array_equal(sel, ref2["selected"])


@pytest.mark.parametrize("gcn_number,gcn_top,precision,selector", [(2, 4, "f32", "fps"), (1, 100, "f32", "fps"), (1, 0, "bf16x3", "fps"), (1, 0, "f32", "kcenter")])
def test_hot_path_selection_variants(backend, gcn_number, gcn_top, precision, selector):
    """gcn_top > 0 (the keep-top mask of fps_gcn_cpu.py:153-160; the reference's scripts run --gcn_top 100), two propagation hops,
    the split-bf16 network arithmetic, and the global k-center selector (BASELINE configuration 4): the selection is exact given the network outputs, the features stay inside 1e-3."""
    from oracle import pipeline_np
    from oracle import randla_np as R
    from ssdr_al import pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS

    class Cfg(ConfigS3DIS):
        pass
    Cfg.num_points = 2048 if backend == "emu" else 40960
    W = R.init_weights(0)
    rooms = [synthetic.make_room(5100 + i, density=150.0 if backend == "emu" else 2500.0) for i in range(2)]
    spt, lpt = (12, 4) if backend == "emu" else (37, 15)
    hp = pipeline.HotPath(W, Cfg, select_per_tile=spt, labeled_per_tile=lpt, gcn_number=gcn_number, gcn_top=gcn_top, precision=precision, selector=selector).load_rooms(rooms)
    sel, unl = hp.step()
    gp, gf = hp.probs.to_host(), hp.f32.to_host()
    ref = pipeline_np.run(hp, rooms, W, threads=4)
    assert np.abs(gp - ref["probs"]).max() < 1e-3 and np.abs(gf - ref["f32"]).max() < 1e-3
    ref2 = pipeline_np.run(hp, rooms, W, threads=4, net_outputs=(gp, gf))
    assert unl == ref2["unl"]
    assert np.array_equal(sel, ref2["selected"])


def test_small_room_is_padded_by_duplication(backend):
    """A room with fewer than num_points sub-sampled points takes the data_aug path (helper_tool.py:185-199)."""
    from oracle import pipeline_np
    hp, rooms, W = _setup(8192 if backend == "emu" else 40960, 1, 60.0 if backend == "emu" else 200.0, 4, 2)
    hp._front_end()
    ref = pipeline_np.run(hp, rooms, W, stop_after="front_end")
    assert ref["m"][0] < hp.cfg.num_points
    assert_bits_equal(hp.xyz.to_host(), ref["xyz"], "padded tile")
    assert_bits_equal(hp.feat.to_host(), ref["feat"], "padded tile features")


_ONE = {}


@pytest.mark.parametrize("depth", [2, 4, 5])
def test_overlapped_batches_give_the_same_selection(backend, depth):
    if backend == "emu" and depth != 5:
        pytest.skip("the CPU logic build runs the default depth only (the stream groupings differ in scheduling, which the stand-in does not model)")
    """bench.py keeps several batches in flight (front end | KNN pyramid | network + scoring | selection on separate
    streams, one buffer set per batch); the result of every batch must not change."""
    from oracle import randla_np as R
    from ssdr_al import pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS

    class Cfg(ConfigS3DIS):
        pass
    Cfg.num_points = 1024 if backend == "emu" else 40960
    W = R.init_weights(0)
    rooms = [synthetic.make_room(5000 + i, density=80.0 if backend == "emu" else 2500.0) for i in range(1 if backend == "emu" else 2)]
    make = lambda: pipeline.HotPath(W, Cfg, select_per_tile=6, labeled_per_tile=3).load_rooms(rooms)
    key = (backend, len(rooms))
    if key not in _ONE:
        _ONE[key] = make().step()[0]
    one = _ONE[key]
    pipe = pipeline.Pipelined(make, depth)
    for k in ((depth + 1,) if backend == "emu" else (1, 2, 5)):       # the emulator is slow: one run that fills and drains the pipe
        sel, _ = pipe.run(k)
        assert np.array_equal(sel, one)
    if backend != "emu" and depth == 5:          # two selections behind the newest in flight (three selection streams, one more buffer set)
        pipe2 = pipeline.Pipelined(make, depth, sel_lag=2)
        for k in (1, 3, 8):
            sel, _ = pipe2.run(k)
            assert np.array_equal(sel, one)
        pipe2.run(4, steady=True); sel, _ = pipe2.run(3, steady=True); pipe2.finish()
        assert np.array_equal(sel, one)
        # the other schedule bench.py offers (--schedule batch, measured slower: profiles/r06_bench_schedules.txt): a batch per stream
        for ss in (0, 2):
            pipe3 = pipeline.BatchStreams(make, 3, ss)
            for k in (1, 7):
                sel, _ = pipe3.run(k)
                assert np.array_equal(sel, one)
            pipe3.finish()


def test_semantic3d_configuration_matches_oracle(backend):
    """The other dataset flavour of the reference (helper_tool.py:77-117): 8 classes, 0.06 m grid, 65536-point tiles (a small
    tile on the CPU logic build); every stage against the oracle like the S3DIS case."""
    from oracle import pipeline_np, randla_np as R
    from ssdr_al import pipeline, synthetic
    from ssdr_al.helper_tool import ConfigSemantic3D

    class Cfg(ConfigSemantic3D):
        pass
    Cfg.num_points = 1024 if backend == "emu" else 65536
    W = R.init_weights(0, num_classes=Cfg.num_classes)
    rooms = [synthetic.make_room(6100 + i, density=60.0 if backend == "emu" else 3000.0) for i in range(2)]
    rooms = [(r[0], r[1], (r[2] % Cfg.num_classes).astype(r[2].dtype)) + tuple(r[3:]) for r in rooms]
    hp = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2).load_rooms(rooms)
    sel, unl = hp.step()
    ref = pipeline_np.run(hp, rooms, W, threads=2)
    assert_bits_equal(hp.xyz.to_host(), ref["xyz"], "tiles")
    for l in range(Cfg.num_layers):
        assert_bits_equal(hp.neigh[l].to_host(), ref["neigh"][l], "neigh level %d" % l)
        assert_bits_equal(hp.interp[l].to_host(), ref["interp"][l], "interp level %d" % l)
    gp, gf = hp.probs.to_host(), hp.f32.to_host()
    assert gp.shape[1] == 8 and np.abs(gp - ref["probs"]).max() < 1e-3 and np.abs(gf - ref["f32"]).max() < 1e-3     # north_star tolerance (fp32)
    ref2 = pipeline_np.run(hp, rooms, W, threads=2, net_outputs=(gp, gf))                # selection: exact given the same network outputs
    assert _ranking_matches(hp, ref2) and unl == ref2["unl"] and np.array_equal(sel, ref2["selected"])


@pytest.mark.parametrize("case", ["cloud_all_labelled", "batch_exceeds_regions", "kcenter", "many_small_regions"])
def test_candidate_rule_on_device_equals_host_rule(backend, case, monkeypatch):
    """sampler2.py:533-552, :745-753 as device kernels (ssdr_gcn_fps_sampling_dev: counts stay on the device) against the vectorised host rule +
    the separate entry points: same candidates, same picks — also when a cloud has no region left to offer and when the batch asks for
    more regions than are unlabelled."""
    if case == "many_small_regions":       # a partition of thousands of regions per cloud: the rule's place-inside-the-cloud counting runs in slices over several workgroups
        from ssdr_al import synthetic
        orig = synthetic.superpoints_from_tile
        monkeypatch.setattr(synthetic, "superpoints_from_tile", lambda xyz, cell=0.3: orig(xyz, 0.05 if backend == "emu" else 0.07))
    if backend == "emu":
        hp, rooms, W = _setup(2048, 2, 150.0, 6, 3)
    else:
        hp, rooms, W = _setup(40960, 3, 2500.0, 37, 15)
    if case == "many_small_regions":
        assert hp.S / hp.B > 1024, hp.S
    if case == "cloud_all_labelled":
        hp.labeled[1] = set(np.flatnonzero(hp.sp_cloud_h == 1).tolist())
    if case == "batch_exceeds_regions":
        hp.select_per_tile = hp.S
    if case == "kcenter":
        hp.selector = "kcenter"          # kCenterGreedy over candidates + labelled rows, seeded with the labelled ones
    hp.set_labeled(hp.labeled)
    monkeypatch.setenv("SSDR_SELECT_HOST_RULE", "1")
    sel_h, unl_h = hp.step()
    picked_h = list(hp.selected)
    monkeypatch.delenv("SSDR_SELECT_HOST_RULE")
    sel_d, unl_d = hp.step()
    assert hp._sel_static["d_result"].to_host()[5] == 0
    assert unl_d == unl_h
    assert np.array_equal(sel_d, sel_h)
    assert hp.selected == picked_h
    if case == "cloud_all_labelled":
        assert all(b != 1 for b, _ in unl_d)
    if case == "batch_exceeds_regions":
        assert len(sel_d) == int((~hp.labeled_mask).sum()) == len(unl_d)


def test_candidate_rule_reports_capacity_overflow(backend):
    """capacities below what the rule produces: nothing is selected and the status says so (no write past the caller's buffers)"""
    hp, rooms, W = _setup(2048, 2, 150.0, 6, 3) if backend == "emu" else _setup(40960, 2, 2500.0, 37, 15)
    hp._sel_static["cap_sq"] = 4
    with pytest.raises(RuntimeError, match="capacities"):
        hp.step()
