"""One active-learning round at the reference's own scale (ssdr_main_S3DIS2.py:134: ONE selection of batch_size = 10 000 regions over
2 x 10 000 candidates + up to 4 000 labelled rows of all 272 rooms; sampler2.py:736-781, fps_gcn_cpu.py:119-178).

Until round 6 the one-call device chain (ssdr_gcn_fps_sampling_dev) refused more than 16 384 rows and 20 000 x 10 000 FPS lived in a tool.
Here: (not gpu / emu) the chain beyond 16 384 rows on the CPU logic build; (gpu) 272 fabricated clouds at the reference's sizes — population,
ranking, candidate list and budget == oracle/pipeline_np, the propagated rows of a sample of clouds == the oracle's float64 graph, the FPS
sequence == the NumPy oracle over the device's rows, device rule == host rule; FPS 20 000 x 10 000 on its own."""
import numpy as np
import pytest

from _fabricate import make_clouds

ARGS = ("sb", "WetSU", "clsbal", "gcn_fps")


def _device_rows(hp, n):
    import ctypes as C
    from ssdr_al import _lib
    p, cap = C.c_void_p(), C.c_size_t()
    _lib.check(_lib.lib().ssdr_gcn_fps_sampling_rows(hp.sel_stream, C.byref(p), C.byref(cap)))
    assert cap.value >= n
    out = np.empty((n, 32), np.float64)
    _lib.check(_lib.lib().ssdr_memcpy_d2h(_lib.ptr(out), p.value, out.nbytes))
    return out


def _round(clouds, labelled, sel_list, C, batch_size, round_num, selector, label_seed, graph_clouds, monkeypatch, host_rule=True):
    from oracle import pipeline_np as P
    from oracle import select_np as S
    from ssdr_al import pipeline
    from ssdr_al.helper_tool import ConfigS3DIS

    class Cfg(ConfigS3DIS):
        num_classes = C
    hp = pipeline.HotPath.from_clouds(clouds, labelled, sel_list, Cfg, sampler_args=ARGS, gcn_number=1, gcn_top=0, min_size=1, round_num=round_num,
                                      label_seed=label_seed, batch_size=batch_size, selector=selector)
    sel, unl = hp.step_selection()
    assert hp.rule_path == "device"
    T = hp._sel_static
    assert T["cap_rows"] > 16384
    r = P.selection_round(clouds, labelled, sel_list, C, list(ARGS), 1, round_num, batch_size, 1, 0, 0, np.random.RandomState(label_seed), selector=selector,
                          graph_clouds=set(graph_clouds))
    base = np.asarray(hp.sp_base)
    # population, ranking, candidates, budget: the oracle's
    pop = np.array([base[b] + s for b, s in r["region"]])
    assert np.array_equal(np.flatnonzero(~hp.skip_mask), pop)
    order = hp.sorted_inds.to_host()
    assert np.array_equal(order[~hp.skip_mask[order]], pop[r["sorted_inds"]])
    assert [(b, s - int(base[b])) for b, s in unl] == r["unl"]
    assert len(sel) == r["sampling_batch"] == batch_size
    rows_lab = sorted((b, s - int(base[b])) for b in hp.lab_rows for s in hp.lab_rows[b])
    assert rows_lab == r["lab"]
    n_unl, n_all = len(unl), len(unl) + len(rows_lab)
    # the propagated rows: a sample of clouds against the oracle's float64 graph (1e-12 relative: libm exp, summation order of the matmul)
    comb = _device_rows(hp, n_all)
    gr = r["graph_rows"]
    assert len(gr) > 0
    assert np.allclose(comb[gr], r["comb"][gr], rtol=1e-11, atol=1e-13)
    # the chain over the device's rows: index for index the NumPy oracle
    if selector == "kcenter":
        exp = S.kcenter_greedy(comb, np.arange(n_unl, n_all), batch_size)
    else:
        exp = S.farthest_features_sample(comb[:n_unl], batch_size, 0)
    assert np.array_equal(sel, np.asarray(exp, np.int32))
    if not host_rule:
        return hp
    # device rule == host rule over the same kernels
    monkeypatch.setenv("SSDR_SELECT_HOST_RULE", "1")
    sel_h, unl_h = hp.step_selection()
    assert hp.rule_path == "host" and unl_h == unl and np.array_equal(sel_h, sel)
    return hp


def test_one_call_chain_beyond_16384_rows(backend, monkeypatch):
    """many labelled rows, few picks: 17 000 + rows through the one-call chain (CPU logic build and GPU)"""
    clouds, labelled, sel_list = make_clouds(11, 420, 52, 2, 3, labelled_per_cloud=41)      # 21 840 regions, 17 220 of them labelled
    _round(clouds, labelled, sel_list, 13, 300, 18, "fps", 5, [0, 7, 419], monkeypatch, host_rule=backend == "gpu")      # (the CPU logic build runs the chain once: minutes otherwise)


@pytest.mark.gpu
@pytest.mark.parametrize("selector", ["fps", "kcenter"])
def test_al_round_at_reference_scale(backend, selector, monkeypatch):
    """272 clouds, batch_size 10 000, 20 000 candidates + 4 000 labelled rows in ONE call of the device chain"""
    if backend != "gpu":
        pytest.skip("the reference's scale runs on the GPU only")
    clouds, labelled, sel_list = make_clouds(3, 272, 150, 20, 60, labelled_per_cloud=15)    # 40 800 regions, 4 080 labelled, 1.6 M points
    hp = _round(clouds, labelled, sel_list, 13, 10000, 5, selector, 9, [0, 100, 271], monkeypatch)
    T = hp._sel_static
    assert T["picks"] == 10000 and T["cap_unl"] == 20000 and T["n_lab"] == 4000


@pytest.mark.gpu
def test_fps_20000_rows_10000_picks(backend):
    """farthest_features_sample at the reference's scale (fps_gcn_cpu.py:119-147, 10 000 picks over 20 000 x 32): the cooperative chain's sequence ==
    the NumPy oracle's (tools/fps_large.py of rounds 3-5 as a test)"""
    if backend != "gpu":
        pytest.skip("the cooperative chain exists on the GPU only")
    from oracle import select_np as S
    from ssdr_al import sampler
    f = np.random.default_rng(1).normal(size=(20000, 32))
    got = sampler.farthest_features_sample(f, 10000, 0)
    assert np.array_equal(got, S.farthest_features_sample(f, 10000, 0))


def test_al_round_plumbing_equals_per_batch_runs(backend):
    """pipeline.ALRound (batches overlapped on streams, tiles / labels / network outputs written into slices of the round's arrays, ONE selection over all
    clouds): every batch's tiles and network outputs == a HotPath of its own over the same rooms and room ids, and the round's selection == the
    selection half run over host copies of the round's arrays (HotPath.from_clouds)"""
    from oracle import randla_np as R
    from ssdr_al import pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS
    emu = backend == "emu"

    class Cfg(ConfigS3DIS):
        num_points = 512 if emu else 40960
    W = R.init_weights(0)
    rooms = [synthetic.make_room(8100 + i, density=70.0 if emu else 2500.0) for i in range(2)]
    nb = 2 if emu else 6          # (the GPU run goes around the four buffer sets)
    ar = pipeline.ALRound(W, rooms, nb, Cfg, batch_size=24, round_num=2, labeled_per_tile=3, precision="f32")
    sel, unl = ar.run()
    sel2, unl2 = ar.run()
    assert np.array_equal(sel, sel2) and unl == unl2 and len(sel) == 24
    N, B = Cfg.num_points, len(rooms)
    xyz, probs, f32, lab = ar.xyz.to_host(), ar.probs.to_host(), ar.f32.to_host(), ar.tile_l.to_host()
    for b in ((nb - 1,) if emu else (0, nb - 1)):
        hp = pipeline.HotPath(W, Cfg, precision="f32").load_rooms(rooms, [b * B + i for i in range(B)])
        hp._front_end(); hp._pyramid(); hp._infer()
        from ssdr_al import _lib
        _lib.sync()
        lo, hi = b * B * N, (b + 1) * B * N
        assert np.array_equal(hp.xyz.to_host().reshape(-1, 3), xyz[lo:hi]) and np.array_equal(hp.tile_l.to_host(), lab[lo:hi])
        assert np.array_equal(hp.probs.to_host(), probs[lo:hi]) and np.array_equal(hp.f32.to_host(), f32[lo:hi])
    # the selection half over host copies of the same arrays
    S = ar.sel
    T = nb * B
    clouds, labelled = [], []
    for t in range(T):
        s0, s1 = S.sp_base[t], (S.sp_base[t + 1] if t + 1 < T else S.S)
        off = S.sp_off_h[s0:s1 + 1].astype(np.int64)
        clouds.append(dict(xyz=xyz[t * N:(t + 1) * N], gt=lab[t * N:(t + 1) * N], probs=probs[t * N:(t + 1) * N], feat=f32[t * N:(t + 1) * N],
                           offsets=off - off[0], points=S.sp_pts_h[off[0]:off[-1]].astype(np.int64) - t * N))
        labelled.append(set(int(x) - s0 for x in S.labeled[t]))
    ref = pipeline.HotPath.from_clouds(clouds, labelled, S.selected_class_list.to_host(), Cfg, batch_size=24, round_num=2)
    rsel, runl = ref.step_selection()
    assert runl == unl and np.array_equal(rsel, sel)


@pytest.mark.gpu
def test_al_round_stage_schedule_and_events(backend, monkeypatch):
    """the other schedule of ALRound.infer_all (a stream per stage, producer waits, an ssdr_event_* for the buffer set's last reader; measured slower and kept for
    A/B runs) selects what the default schedule selects; the event entry points refuse NULL"""
    if backend != "gpu":
        pytest.skip("timing variant: GPU only")
    import ctypes as C
    from oracle import randla_np as R
    from ssdr_al import _lib, pipeline, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS
    L = _lib.lib()
    assert L.ssdr_event_record(None, None) != 0 and L.ssdr_stream_wait_event(None, None) != 0 and L.ssdr_event_create(None) != 0
    ev = C.c_void_p()
    _lib.check(L.ssdr_event_create(C.byref(ev))); _lib.check(L.ssdr_event_record(ev, None)); _lib.check(L.ssdr_stream_wait_event(None, ev)); _lib.check(L.ssdr_event_destroy(ev))

    class Cfg(ConfigS3DIS):
        num_points = 8192
    W = R.init_weights(0)
    rooms = [synthetic.make_room(8200 + i, density=800.0) for i in range(2)]
    ar = pipeline.ALRound(W, rooms, 7, Cfg, batch_size=40, round_num=2, labeled_per_tile=3, precision="f32")
    sel, unl = ar.run()
    monkeypatch.setenv("SSDR_AL_SCHED", "stage")
    sel2, unl2 = ar.run()
    assert unl2 == unl and np.array_equal(sel2, sel) and len(sel) == 40
