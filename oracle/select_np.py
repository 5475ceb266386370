"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the reference's selection-stage numerics.

Pinned against the reference's own Python (imported from /root/reference in the build container by
tests/golden/make_golden_select.py; vectors in tests/golden/select_golden.npz).  Paths relative to
/root/reference/SSDR_AL_s3dis/.

Data layout used here instead of the reference's pickled object arrays: superpoints of one cloud are a CSR pair
(offsets int32 [S+1], points int32 [T]) — `components[s]` == points[offsets[s]:offsets[s+1]]
(partition/compute_superpoint.py:63-68).
"""
import numpy as np


# ---- sampler2.py:28-47 ------------------------------------------------------------------------------------
def point_uncertainty(prob, mode):
    prob = np.asarray(prob)
    if mode == "lc":
        return 1.0 - np.max(prob, axis=-1)
    if mode == "entropy":                                  # compute_entropy :247-255
        with np.errstate(divide="ignore"):
            k = np.log2(prob)
        k[np.isinf(k)] = 0
        return -1 * np.sum(np.multiply(prob, k), axis=-1)
    if mode == "sb":
        s = np.sort(prob, axis=-1)
        return s[:, -2] / s[:, -1]
    raise ValueError(mode)


def dominant_label(ary):                                   # sampler2.py:102-106
    h = np.bincount(np.asarray(ary), minlength=1)
    return int(np.argmax(h)), h.max() / len(ary)


def region_uncertainty(pu, pc, class_num, mode):           # sampler2.py:12-26
    if mode == "mean":
        return np.mean(pu)
    if mode == "sum_weight":                               # weights_percentage :92-100
        h = np.bincount(pc, minlength=class_num) / len(pc)
        return np.sum(np.multiply(h[pc], pu))
    if mode == "WetSU":
        d, _ = dominant_label(pc)
        eq = np.where(pc == d, 1.0, 0.0)
        return np.sum(np.multiply(pu, eq)) - np.sum(np.multiply(pu, 1 - eq))
    raise ValueError(mode)


def region_stats(unc, cls, offsets, points, class_num, mode, min_size=0):
    """The per-superpoint loop of TSampler.prediction (sampler2.py:612-626)."""
    S = len(offsets) - 1
    ru = np.zeros(S, np.float64); dom = np.zeros(S, np.int32); cnt = np.zeros(S, np.int32)
    for s in range(S):
        ids = points[offsets[s]:offsets[s + 1]]
        if len(ids) == 0:
            continue
        ru[s] = region_uncertainty(unc[ids], cls[ids], class_num, mode)
        d, _ = dominant_label(cls[ids])
        dom[s] = d
        cnt[s] = int((cls[ids] == d).sum())
    return ru, dom, cnt


def add_clsbal(class_num, region_class, region_unc, selected_class_list=()):   # sampler2.py:262-266
    lst = list(region_class) + list(selected_class_list)
    h = np.bincount(np.asarray(lst, np.int64), minlength=class_num) / len(lst)
    w = h[np.asarray(region_class, np.int64)]
    return np.multiply(region_unc, np.exp(-w))


def rank_regions(region_unc):
    """sorted_inds = argsort(-u) (sampler2.py:640).  numpy's default sort is not stable, so the order among
    *equal* uncertainties is unspecified in the reference; here (and in the HIP path) ties go by ascending index."""
    return np.argsort(-np.asarray(region_unc), kind="stable")


def segment_mean_features(feat, offsets, points, cls, dom):
    """compute_features (sampler2.py:333,339): mean of last_second_features over the dominant-class members."""
    S = len(offsets) - 1
    out = np.zeros((S, feat.shape[1]), np.float32)
    for s in range(S):
        ids = points[offsets[s]:offsets[s + 1]]
        ids = ids[cls[ids] == dom[s]]
        out[s] = np.mean(feat[ids], axis=0)
    return out


# ---- fps_gcn_cpu.py ------------------------------------------------------------------------------------------
def bbox_centres(xyz, offsets, points):                    # fps_gcn_cpu.py:86-88
    S = len(offsets) - 1
    c = np.zeros((S, 3))
    for s in range(S):
        p = xyz[points[offsets[s]:offsets[s + 1]]]
        # float32 min + float32 max is a float32 add in the reference (under NumPy 1.16 and 2.x alike); the
        # halving is exact, so the centre is float32(min + max) / 2 stored in a float64 array.
        c[s] = (np.min(p, 0) + np.max(p, 0)).astype(np.float32).astype(np.float64) / 2.0
    return c


def create_cd(xyz, offsets, points, centres):
    """create_cd / chamfer_distance (fps_gcn_cpu.py:12-38) with brute-force nearest neighbours (sklearn's
    KDTree.query is exact): cd[i,j] = mean_{a in j} min_{b in i} |a-b| + mean_{b in i} min_{a in j} |a-b|."""
    S = len(offsets) - 1
    al = [xyz[points[offsets[s]:offsets[s + 1]]] - centres[s] for s in range(S)]
    dirm = np.zeros((S, S))                                # dirm[i,j] = mean over points of i of NN distance into j
    for i in range(S):
        for j in range(S):
            if i != j:
                d = al[i][:, None, :] - al[j][None, :, :]
                dist = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
                dirm[i, j] = np.mean(dist.min(1))
    return dirm + dirm.T


def create_cd_cuda(xyz, offsets, points, centres):
    """create_cd_cuda (SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30), the Semantic3D flavour: the centred superpoints are rounded to float32 (torch.Tensor),
    chamfer3D.cu gives squared float32 nearest distances ((dx*dx + dy*dy) + dz*dz, dx = b - a), and cd[i,j] = mean(sqrt(dist1)) + mean(sqrt(dist2)) in
    float32, widened.  PARITY UNPINNED: the CUDA op cannot run here, and torch.mean's reduction order on CUDA is the library's — compare at 1e-6 relative."""
    S = len(offsets) - 1
    al = [(xyz[points[offsets[s]:offsets[s + 1]]].astype(np.float64) - centres[s]).astype(np.float32) for s in range(S)]
    dirm = np.zeros((S, S), np.float32)
    for i in range(S):
        for j in range(S):
            if i != j and len(al[i]) and len(al[j]):
                d = al[j][None, :, :] - al[i][:, None, :]
                dist = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
                dirm[i, j] = np.mean(np.sqrt(dist.min(1)), dtype=np.float32)
    return (dirm + dirm.T).astype(np.float64)


def block_adjacency(centres, cd):
    """One cloud's block of fps_adj_all (fps_gcn_cpu.py:95-115): exp(-(ED+CD)), minus I, column-scaled by the
    inverse row sums, plus I.  Entries across clouds are exactly 0 (exp(-2e10)), so blocks are independent."""
    d = centres[:, None, :] - centres[None, :, :]
    ed = np.sqrt(np.sum(d * d, -1))
    adj = np.exp(-(ed + cd))
    adj = adj - np.eye(len(adj))
    rs = adj.sum(1)
    with np.errstate(divide="ignore"):
        dinv = np.power(rs, -1)
    dinv[np.isinf(dinv)] = 0.0
    return adj * dinv[None, :] + np.eye(len(adj))


def create_adj(features, centres_blocks, cd_blocks, block_rows):
    """gcn.create_adj (gcn.py:116-191), the adjacency of the trained-GCN branch, restated in NumPy float32 (the reference runs it as torch
    float32): rows L2-normalised (torch.nn.functional.normalize, eps 1e-12), cosine matrix, times exp(-(ED + CD)) — both float64 matrices
    cast to float32 first, 1e10 between clouds so those entries are exactly 0 —, minus I, columns scaled by the inverse column sums, plus I.
    centres_blocks / cd_blocks: per cloud the bbox centres [n_c,3] (float64) and chamfer matrix [n_c,n_c] (float64, create_cd);
    block_rows: per cloud the row of every member in the [N, N] result."""
    f = np.asarray(features, np.float32)
    nrm = np.sqrt(np.sum(f * f, axis=1, dtype=np.float32)).astype(np.float32)
    V = (f / np.maximum(nrm, np.float32(1e-12))[:, None]).astype(np.float32)
    N = len(V)
    a_ed = np.ones((N, N)) * 1e10; a_cd = np.ones((N, N)) * 1e10
    for cen, cd, rows in zip(centres_blocks, cd_blocks, block_rows):
        d = cen[:, None, :] - cen[None, :, :]
        a_ed[np.ix_(rows, rows)] = np.sqrt(np.sum(np.multiply(d, d), axis=-1))
        a_cd[np.ix_(rows, rows)] = cd
    lat = (V @ V.T).astype(np.float32)
    adj = (lat * np.exp(-(a_ed.astype(np.float32) + a_cd.astype(np.float32)))).astype(np.float32)
    adj = adj - np.eye(N, dtype=np.float32)
    diag = adj.sum(0, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        adj = (adj * (np.float32(1.0) / diag)[None, :]).astype(np.float32)
    return V, adj + np.eye(N, dtype=np.float32)


def keep_top(adj, gcn_top):
    """The keep-top mask of GCN_FPS_sampling (fps_gcn_cpu.py:153-160) on one cloud's block: every row keeps its gcn_top largest
    entries.  The reference sorts whole rows of the global matrix; entries outside the block are exactly 0 and block entries are
    positive, so the row's top entries are the block's (and a mask on zeros changes nothing)."""
    gcn_top = int(gcn_top)
    if gcn_top <= 0 or gcn_top >= adj.shape[1]:
        return adj
    out = np.zeros_like(adj)
    for i in range(adj.shape[0]):
        keep = np.argsort(adj[i], kind="stable")[-gcn_top:]
        out[i, keep] = adj[i, keep]
    return out


def propagate(adj_blocks, block_rows, V, gcn_number):
    """sum_{i=0..gcn_number} A^i V (fps_gcn_cpu.py:162-167), A block-diagonal."""
    V = np.asarray(V, np.float64)
    total = V.copy(); cur = V.copy()
    for _ in range(int(gcn_number)):
        nxt = np.zeros_like(cur)
        for A, rows in zip(adj_blocks, block_rows):
            nxt[rows] = A @ cur[rows]
        cur = nxt
        total = total + cur
    return total


def farthest_features_sample(features, sample_number, start):   # fps_gcn_cpu.py:119-147
    f = np.asarray(features, np.float64)
    cent = np.zeros(sample_number, np.int32)
    cent[0] = start
    distance = np.ones(len(f)) * 1e10
    for i in range(sample_number - 1):
        dist = np.sum((f - f[cent[i]]) ** 2, axis=-1)
        mask = dist < distance
        distance[mask] = dist[mask]
        cent[i + 1] = np.argmax(distance)
    return cent


def farthest_superpoint_sample(xyz, offsets, points, sel, sample_number, trigger_idx):
    """sampler2.py:49-80 (the "edcd" branch): FPS over superpoints with distance |centre_i - centre_c|^2 + CD(i, c)."""
    sub_off = np.concatenate([[0], np.cumsum(offsets[np.asarray(sel) + 1] - offsets[np.asarray(sel)])]).astype(np.int32)
    sub_pts = np.concatenate([points[offsets[s]:offsets[s + 1]] for s in sel])
    cen = bbox_centres(xyz, sub_off, sub_pts)
    cd = create_cd(xyz, sub_off, sub_pts, cen)
    n = len(sel)
    cent = np.zeros(sample_number, np.int32); cent[0] = trigger_idx
    distance = np.ones(n) * 1e10
    for i in range(sample_number - 1):
        dist = np.sum((cen - cen[cent[i]]) ** 2, axis=-1) + cd[cent[i]]
        mask = dist < distance
        distance[mask] = dist[mask]
        cent[i + 1] = np.argmax(distance)
    return cent


def kcenter_greedy(features, already_selected, n):          # kcenterGreedy.py:60-128 (direct distances)
    f = np.asarray(features, np.float64)
    md = None
    if len(already_selected):
        d = np.sqrt(((f[:, None, :] - f[None, np.asarray(already_selected), :]) ** 2).sum(-1))
        md = d.min(1)
    out = []
    for _ in range(n):
        ind = int(np.argmax(md)) if md is not None else 0
        d = np.sqrt(((f - f[ind]) ** 2).sum(-1))
        md = d if md is None else np.minimum(md, d)
        out.append(ind)
    return np.asarray(out, np.int32)


def dominant_labels(labels, offsets, points):               # _dominant_label over oracle_labeling "dominant" (:127-144)
    S = len(offsets) - 1
    lab = np.zeros(S, np.int32); purity = np.zeros(S)
    for s in range(S):
        ids = points[offsets[s]:offsets[s + 1]]
        if len(ids):
            lab[s], purity[s] = dominant_label(labels[ids])
    return lab, purity
