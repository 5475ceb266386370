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
    return txyz.astype(np.float32), feat.astype(np.float32), len(sp), sl.reshape(-1)[ids].astype(np.int32)      # queried_pc_label (:141)


def weights_percentage(list_class, class_num):
    """sampler2.py:92-100 as written."""
    dist = np.zeros([class_num])
    for c in list_class:
        dist[c] = dist[c] + 1
    dist = dist / len(list_class)
    return np.asarray([dist[c] for c in list_class])


def labelled_selection(clouds, labelled, class_num, round_num, random_state):
    """get_labeled_selection_cloudname_spidx_pointidx (sampler2.py:268-311): every labelled region's dominant GROUND-TRUTH label and the
    ids of its points that carry it (_dominant_2 over cloud_point_label[point_ids], :288-291), then the class-balanced draw of
    (round_num - 1) * 1000 of them without replacement (:294-302).  clouds[b] = dict(gt, offsets, points, ...); labelled[b] = the labelled
    regions of cloud b that passed min_size, in the order prediction() met them (ascending).  Returns [(cloud, sp, dominant ids)] in draw order."""
    refs, doms = [], []
    for b, cl in enumerate(clouds):
        for s in labelled[b]:
            ids = np.asarray(cl["points"][cl["offsets"][s]:cl["offsets"][s + 1]])
            lab = np.asarray(cl["gt"])[ids].astype(np.int64)
            d = int(np.argmax(np.bincount(lab)))
            refs.append((b, int(s), ids[lab == d])); doms.append(d)
    if not refs:
        return []
    w = weights_percentage(doms, class_num)
    prob = w / np.sum(w)
    batch = min((round_num - 1) * 1000, len(refs))
    selection = random_state.choice(a=len(refs), size=batch, replace=False, p=prob)
    return [refs[i] for i in selection]


def selection_round(clouds, labelled, selected_class_list, class_num, sampler_args, min_size, round_num, batch_size, gcn_number, gcn_top,
                    start, random_state, selector="fps", max_size=None, graph_clouds=None, chamfer="f64"):
    """TSampler.sampling, gcn_fps branch (sampler2.py:736-781), over in-memory clouds instead of files.  clouds[b] = dict(xyz [n,3] f32,
    gt [n] int, probs [n,C] f32, feat [n,32] f32, offsets [S+1], points [T]); labelled[b] = ids of the regions NOT in total_obj["unlabeled"].

    prediction() (:580-642): the population is the UNLABELLED regions of at least min_size points, cloud by cloud — only they get an
    uncertainty, a dominant predicted class and a place in region_class, so add_clsbal's histogram and its length are over them (+ the
    already-selected list); the labelled regions of at least min_size points go to labeled_region_reference_dict.
    create_file_top_and_all (:533-552) + :745-753: the first batch_size ranked regions fix selected_num per cloud, each cloud offers its first
    2 x selected_num ranked regions.  compute_features (:313-342): candidates' means over the dominant PREDICTED class members, labelled
    regions' over the dominant GROUND-TRUTH class members.  Orders the reference leaves to a shuffled DataLoader / a random draw are
    canonical here: candidates cloud ascending, descending uncertainty inside a cloud; labelled rows cloud ascending, superpoint ascending.

    graph_clouds (checks at the reference's scale, 272 clouds: the float64 chamfer loops of every cloud take minutes here): only these clouds' graphs are
    built; `comb` is then valid for their rows alone (`graph_rows`), no sequence is drawn (`seq` None) — the caller runs the FPS oracle over the rows it has."""
    um = [a for a in sampler_args if a in ("lc", "entropy", "sb")][0]
    rm = [a for a in sampler_args if a in ("mean", "sum_weight", "WetSU")][0]
    ref, ru, rclass, lab_ge, cls_of = [], [], [], [], []
    for b, cl in enumerate(clouds):
        probs = np.asarray(cl["probs"])
        cls = np.argmax(probs, axis=-1); unc = S.point_uncertainty(probs, um)
        cls_of.append(cls)
        lab_ge.append([])
        off, pts = cl["offsets"], cl["points"]
        for s in range(len(off) - 1):
            ids = pts[off[s]:off[s + 1]]
            if len(ids) < min_size or (max_size is not None and len(ids) > max_size):      # (the Semantic3D flavour: len <= 1000, S3D/sampler2.py:644, :655)
                continue
            if s in labelled[b]:
                lab_ge[b].append(s)
                continue
            ru.append(S.region_uncertainty(unc[ids], cls[ids], class_num, rm))
            d, _ = S.dominant_label(cls[ids])
            ref.append((b, s)); rclass.append(d)
    raw = np.asarray(ru, np.float64)
    if "classbal" in sampler_args:
        ru = S.add_clsbal(class_num, rclass, raw, ())                              # add_classbal :256-260
    elif "clsbal" in sampler_args:
        ru = S.add_clsbal(class_num, rclass, raw, selected_class_list)
    else:
        ru = raw
    sorted_inds = S.rank_regions(ru)
    t_rank = time.perf_counter()
    batch_size = min(batch_size, len(ref))
    top, allr = {}, {}
    for i, idx in enumerate(sorted_inds):
        b, s = ref[idx]
        if i < batch_size:
            top.setdefault(b, []).append(s)
        allr.setdefault(b, []).append(s)
    labsel = labelled_selection(clouds, lab_ge, class_num, round_num, random_state)
    sampling_batch = sum(len(v) for v in top.values())
    unl = [(b, s) for b in sorted(top) for s in allr[b][: 2 * len(top[b])]]
    lab = sorted((b, s) for b, s, _ in labsel)
    gt_ids = {(b, s): ids for b, s, ids in labsel}
    uf, lf = [], []
    for b, s in unl:
        cl = clouds[b]; ids = cl["points"][cl["offsets"][s]:cl["offsets"][s + 1]]
        d, _ = S.dominant_label(cls_of[b][ids])
        uf.append(np.mean(np.asarray(cl["feat"])[ids[cls_of[b][ids] == d]], axis=0))
    for b, s in lab:
        lf.append(np.mean(np.asarray(clouds[b]["feat"])[gt_ids[(b, s)]], axis=0))
    refs = unl + lab
    nf = np.asarray(clouds[0]["feat"]).shape[1]
    V = np.concatenate([np.asarray(uf, np.float32).reshape(len(unl), nf), np.asarray(lf, np.float32).reshape(len(lab), nf)]).astype(np.float64)
    blocks, rows_l = [], []
    ref_cloud = np.array([c for c, _ in refs], np.int64)
    for b in sorted(set(c for c, _ in refs)):
        if graph_clouds is not None and b not in graph_clouds:
            continue
        rows = np.flatnonzero(ref_cloud == b)
        cl = clouds[b]
        so = np.concatenate([[0], np.cumsum([cl["offsets"][refs[i][1] + 1] - cl["offsets"][refs[i][1]] for i in rows])]).astype(np.int32)
        spts = np.concatenate([cl["points"][cl["offsets"][refs[i][1]]:cl["offsets"][refs[i][1] + 1]] for i in rows])
        xyz = np.asarray(cl["xyz"], np.float32)
        cen = S.bbox_centres(xyz, so, spts)
        cd = S.create_cd_cuda(xyz, so, spts, cen) if chamfer == "f32_cuda" else S.create_cd(xyz, so, spts, cen)      # (the Semantic3D code's float32 CUDA values, fps_gcn_cuda.py:13-30)
        blocks.append(S.keep_top(S.block_adjacency(cen, cd), gcn_top)); rows_l.append(rows)
    comb = S.propagate(blocks, rows_l, V, gcn_number)
    if graph_clouds is not None:
        return dict(region=ref, region_class=np.asarray(rclass, np.int32), region_unc_raw=raw, region_unc=np.asarray(ru, np.float64), sorted_inds=sorted_inds,
                    labelled_ge_min=lab_ge, labsel=labsel, unl=unl, lab=lab, unl_feat=np.asarray(uf, np.float32), lab_feat=np.asarray(lf, np.float32),
                    sampling_batch=sampling_batch, comb=comb, graph_rows=np.concatenate(rows_l) if rows_l else np.zeros(0, np.int64), seq=None, selected=None)
    if sampling_batch == 0:
        seq = np.zeros(0, np.int32)
    elif selector == "kcenter":      # kCenterGreedy over candidates + labelled rows, the labelled ones already selected (gcn.py:247)
        seq = S.kcenter_greedy(comb, np.arange(len(unl), len(refs)), sampling_batch)
    else:
        seq = S.farthest_features_sample(comb[:len(unl)], sampling_batch, start)
    return dict(region=ref, region_class=np.asarray(rclass, np.int32), region_unc_raw=raw, region_unc=np.asarray(ru, np.float64), sorted_inds=sorted_inds, t_rank=t_rank,
                labelled_ge_min=lab_ge, labsel=labsel, unl=unl, lab=lab, unl_feat=np.asarray(uf, np.float32), lab_feat=np.asarray(lf, np.float32),
                sampling_batch=sampling_batch, comb=comb, seq=np.asarray(seq, np.int32), selected=[unl[i] for i in seq])


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
    # the selection round in the reference's own form (selection_round above), every tile a cloud with its superpoints and tile labels
    labels = np.stack([a[3] for a in tiles])
    clouds, labelled = [], []
    for b in range(hp.B):
        lo, hi = hp.sp_base[b], (hp.sp_base[b + 1] if b + 1 < hp.B else hp.S)
        off = hp.sp_off_h[lo:hi + 1] - hp.sp_off_h[lo]
        pts = hp.sp_pts_h[hp.sp_off_h[lo]:hp.sp_off_h[hi]] - b * N
        clouds.append(dict(xyz=xyz0[b], gt=labels[b], probs=probs[b * N:(b + 1) * N], feat=f32[b * N:(b + 1) * N], offsets=off, points=pts))
        labelled.append(set(int(s) - lo for s in hp.labeled.get(b, ())))
    r = selection_round(clouds, labelled, hp.selected_class_list.to_host(), cfg.num_classes, hp.sampler_args, hp.min_size, hp.round_num,
                        hp._sel_static["batch"], hp.gcn_number, hp.gcn_top, 0, np.random.RandomState(hp.label_seed), getattr(hp, "selector", "fps"), getattr(hp, "max_size", None))
    t += [r["t_rank"], time.perf_counter()]
    um = [a for a in hp.sampler_args if a in ("lc", "entropy", "sb")][0]
    out.update(unc=S.point_uncertainty(probs, um), cls=np.argmax(probs, -1).astype(np.int32))
    base = np.asarray(hp.sp_base, np.int64)
    # region: the ranked population as (cloud, global superpoint id); ranked = those ids in rank order (what the product's ranking of all
    # regions reads once the regions outside the population are dropped)
    region = [(b, s + int(base[b])) for b, s in r["region"]]
    out.update(labels=labels, region_unc=r["region_unc"], region=region, sorted_inds=r["sorted_inds"], ranked=np.array([region[i][1] for i in r["sorted_inds"]], np.int64),
               selected=r["seq"], unl=[(b, s + int(base[b])) for b, s in r["unl"]], lab=[(b, s + int(base[b])) for b, s in r["lab"]], comb=r["comb"],
               stage_ms=dict(zip(("subsample+tile", "knn_pyramid", "randla_infer", "score", "select"), np.diff(t) * 1e3)))
    return out
