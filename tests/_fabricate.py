"""Fabricated clouds for the selection half of the path (no front end, no network): what prediction() / compute_features see per cloud
(sampler2.py:580-642, :313-342) — points, ground truth, class probabilities, 32-d features, superpoints — at sizes a test chooses.
Test infrastructure (inputs only; the expected values come from oracle/ or from the golden vectors)."""
import numpy as np


def make_clouds(seed, n_clouds, n_regions, pts_lo, pts_hi, num_classes=13, labelled_per_cloud=15):
    """-> clouds [dict(xyz, gt, probs, feat, offsets, points)], labelled [set of region ids per cloud], selected_class_list"""
    rng = np.random.default_rng(seed)
    clouds, labelled = [], []
    for b in range(n_clouds):
        nsp = int(n_regions if np.isscalar(n_regions) else rng.integers(n_regions[0], n_regions[1] + 1))
        szs = rng.integers(pts_lo, pts_hi + 1, nsp)
        n = int(szs.sum())
        centres = rng.random((nsp, 3)) * np.array([6.0, 5.0, 2.5])
        sp_of = np.repeat(np.arange(nsp), szs)
        xyz = (centres[sp_of] + rng.normal(0, 0.15, (n, 3))).astype(np.float32)
        perm = rng.permutation(n)                      # a region's points are scattered over the cloud, as in the reference's .superpoint files
        xyz = xyz[perm]
        inv = np.argsort(perm)
        offsets = np.concatenate([[0], np.cumsum(szs)]).astype(np.int32)
        points = inv.astype(np.int32)                  # CSR: the ids of region s are points[offsets[s]:offsets[s+1]]
        base = rng.integers(0, num_classes, nsp)
        gt_sorted = np.where(rng.random(n) < 0.7, base[sp_of], rng.integers(0, num_classes, n))
        gt = np.empty(n, np.int32); gt[points] = gt_sorted
        logits = rng.normal(0, 1.0, (n, num_classes)).astype(np.float32)
        pred_sorted = np.where(rng.random(n) < 0.6, base[sp_of], rng.integers(0, num_classes, n))
        logits[points, pred_sorted] += 2.0
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        feat = (rng.normal(0, 1.0, (n, 32)) + 0.5 * rng.normal(0, 1.0, (nsp, 32))[np.argsort(np.argsort(points)) * 0 + _region_of(points, offsets, n)]).astype(np.float32)
        clouds.append(dict(xyz=xyz, gt=gt, probs=probs, feat=feat, offsets=offsets, points=points))
        labelled.append(set(rng.choice(nsp, min(labelled_per_cloud, nsp), replace=False).tolist()))
    return clouds, labelled, rng.integers(0, num_classes, 57).astype(np.int32)


def _region_of(points, offsets, n):
    r = np.empty(n, np.int64)
    r[points] = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets))
    return r
