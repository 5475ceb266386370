"""TEST INFRASTRUCTURE ONLY — the hot path end to end on the CPU from the oracle pieces (oracle/*.c, randla_np.py,
select_np.py), mirroring ssdr_al/pipeline.py stage by stage on the same rooms and the same host-drawn randomness.
Used by the parity tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import time

import numpy as np

from . import c as _c
from . import randla_np as R
from . import select_np as S


def front_end(room, center, perm, dup, num_points, dl):
    """grid-subsample (key order) + spatially_regular_gen (s3dis_dataset.py:115-154) for one room."""
    o = _c()
    xyz, rgb, lab = room
    sp, sc, sl = o.grid_subsampling(xyz, rgb.astype(np.float32), lab.astype(np.int32), dl, order="key")
    c = center.astype(np.float32)
    d = sp - c[None]
    dist = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    order = np.argsort(dist, kind="stable")
    m = len(sp); avail = min(m, num_points)
    if avail == num_points:
        pos = perm
    else:
        shuffled = perm[perm < avail]                       # the shuffle restricted to the rows that exist
        want = np.where(np.arange(num_points) < avail, np.arange(num_points),
                        np.minimum((dup * np.float32(avail)).astype(np.int64), avail - 1))
        pos = shuffled[want]
    ids = order[pos]
    txyz = sp[ids] - c[None]
    feat = np.concatenate([txyz, sc[ids] * np.float32(1.0 / 255.0)], 1)
    return txyz.astype(np.float32), feat.astype(np.float32), len(sp)


def candidates(sorted_inds, labeled, sp_cloud, batch_size, num_clouds):
    """The candidate rule of sampling(), restated with plain loops (sampler2.py:533-552 create_file_top_and_all, :745-753):
    walk the regions by descending uncertainty, skip labelled ones; the first `batch_size` survivors are the "top" regions and
    fix selected_num per cloud; every cloud keeps its first 2 x selected_num survivors as candidates.  Candidate order: cloud
    ascending, descending uncertainty inside a cloud; labelled list: cloud ascending, superpoint ascending."""
    per_cloud = {b: [] for b in range(num_clouds)}
    ntop = {b: 0 for b in range(num_clouds)}
    seen = 0
    batch_size = min(batch_size, len(sorted_inds))
    for s in sorted_inds:
        b = int(sp_cloud[s])
        if int(s) in labeled.get(b, ()):
            continue
        per_cloud[b].append(int(s))
        if seen < batch_size:
            ntop[b] += 1
        seen += 1
    unl = [(b, s) for b in range(num_clouds) for s in per_cloud[b][: 2 * ntop[b]]]
    lab = [(b, s) for b in sorted(labeled) for s in sorted(labeled[b])]
    return unl, lab, sum(ntop.values())


def run(hp, rooms, weights, threads=1, net_outputs=None, stop_after=None):
    """hp: the set-up ssdr_al.pipeline.HotPath (supplies centres, permutations, superpoints, labelled sets)."""
    cfg = hp.cfg
    o = _c()
    N = cfg.num_points
    t = [time.perf_counter()]
    tiles = [front_end(r, hr["center"], hr["perm"], hr["dup"], N, cfg.sub_grid_size) for r, hr in zip(rooms, hp.rooms)]
    xyz0 = np.stack([a[0] for a in tiles]); feat = np.stack([a[1] for a in tiles])
    t.append(time.perf_counter())
    out = {"xyz": xyz0, "feat": feat, "m": [a[2] for a in tiles]}
    if stop_after == "front_end":
        return out
    xyz, neigh, sub, interp = R.build_pyramid(xyz0, cfg.sub_sampling_ratio, lambda s, q, k: o.knn_batch(s, q, k, threads=threads), cfg.k_n)
    out.update(neigh=neigh, interp=interp)
    t.append(time.perf_counter())
    if net_outputs is None:
        probs, f32 = R.forward(weights, feat, xyz, neigh, sub, interp, dtype=np.float32)
    else:
        probs, f32 = net_outputs
    out.update(probs=probs, f32=f32)
    t.append(time.perf_counter())
    um = [a for a in hp.sampler_args if a in ("lc", "entropy", "sb")][0]
    rm = [a for a in hp.sampler_args if a in ("mean", "sum_weight", "WetSU")][0]
    unc = S.point_uncertainty(probs, um); cls = np.argmax(probs, -1).astype(np.int32)
    ru, dom, cnt = S.region_stats(unc, cls, hp.sp_off_h, hp.sp_pts_h, cfg.num_classes, rm)
    if "clsbal" in hp.sampler_args:
        ru = S.add_clsbal(cfg.num_classes, dom, ru, hp.selected_class_list.to_host())
    sorted_inds = S.rank_regions(ru)
    out.update(unc=unc, cls=cls, region_unc=ru, dom=dom, sorted_inds=sorted_inds)
    t.append(time.perf_counter())
    unl, lab, sampling_batch = candidates(sorted_inds, hp.labeled, hp.sp_cloud_h, hp.select_per_tile * hp.B, hp.B)
    refs = unl + lab
    sel = np.array([s for _, s in refs], np.int32)
    sub_off = np.concatenate([[0], np.cumsum(hp.sp_off_h[sel + 1] - hp.sp_off_h[sel])]).astype(np.int32)
    sub_pts = np.concatenate([hp.sp_pts_h[hp.sp_off_h[s]:hp.sp_off_h[s + 1]] for s in sel])
    V = S.segment_mean_features(f32, sub_off, sub_pts, cls, dom[sel]).astype(np.float64)
    flat = xyz0.reshape(-1, 3)
    blocks, rows_l = [], []
    for b in sorted(set(c for c, _ in refs)):
        rows = np.array([i for i, (c, _) in enumerate(refs) if c == b])
        so = np.concatenate([[0], np.cumsum(sub_off[rows + 1] - sub_off[rows])]).astype(np.int32)
        spts = np.concatenate([sub_pts[sub_off[i]:sub_off[i + 1]] for i in rows])
        cen = S.bbox_centres(flat, so, spts)
        blocks.append(S.keep_top(S.block_adjacency(cen, S.create_cd(flat, so, spts, cen)), hp.gcn_top)); rows_l.append(rows)
    comb = S.propagate(blocks, rows_l, V, hp.gcn_number)
    if getattr(hp, "selector", "fps") == "kcenter":      # kCenterGreedy over candidates + labelled rows, the labelled ones already selected (gcn.py:247)
        seq = S.kcenter_greedy(comb, np.arange(len(unl), len(refs)), sampling_batch)
    else:
        seq = S.farthest_features_sample(comb[:len(unl)], sampling_batch, 0)
    t.append(time.perf_counter())
    out.update(selected=seq, unl=unl, comb=comb, stage_ms=dict(zip(("subsample+tile", "knn_pyramid", "randla_infer", "score", "select"), np.diff(t) * 1e3)))
    return out
