"""Host-side mirror of the numeric part of the reference's selection code
(/root/reference/SSDR_AL_s3dis/sampler2.py, fps_gcn_cpu.py, kcenterGreedy.py): same function names and argument
meaning where the reference has a function, arrays instead of pickles/PLY files (file I/O is out of scope).
Every computation runs in libssdr_al.so; this module only moves arrays and sequences the calls.

Superpoints are passed as CSR (offsets int32 [S+1], points int32 [T]); ``csr_from_components`` converts the
reference's ``components`` object array (partition/compute_superpoint.py:63-68)."""
import numpy as np

from . import _lib
from ._lib import DevArray

_UNC = {"lc": 0, "entropy": 1, "sb": 2}
_REG = {"mean": 0, "sum_weight": 1, "WetSU": 2}


def _mode(sampler_args, table):
    for k in table:                       # the reference tests `"lc" in sampler_args` etc. in this order
        if k in sampler_args:
            return table[k]
    raise ValueError("no known mode in %r" % (sampler_args,))


def csr_from_components(components):
    sizes = np.array([len(c) for c in components], np.int64)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pts = np.concatenate([np.asarray(c, np.int32) for c in components]) if len(components) else np.zeros(0, np.int32)
    return off, pts.astype(np.int32)


def compute_point_uncertainty(prob_logits, sampler_args, return_class=False):
    """sampler2.py:28-47 (+ np.argmax of :602 when return_class)."""
    prob = np.ascontiguousarray(prob_logits, np.float32)
    n, C = prob.shape
    d_p = DevArray.from_host(prob); d_u = DevArray((n,), np.float32); d_c = DevArray((n,), np.int32)
    _lib.check(_lib.lib().ssdr_point_uncertainty_dev(d_p.ptr, n, C, _mode(sampler_args, _UNC), d_u.ptr, d_c.ptr, None))
    _lib.sync()
    return (d_u.to_host(), d_c.to_host()) if return_class else d_u.to_host()


def compute_region_stats(pixel_uncertainty, pixel_class, offsets, points, class_num, sampler_args):
    """The per-superpoint loop of TSampler.prediction (sampler2.py:612-626) for one cloud:
    region uncertainty (compute_region_uncertainty :12-26), dominant class (_dominant_label :102-106) and the
    number of dominant-class members (_dominant_2 :108-115)."""
    S = len(offsets) - 1
    d_u = DevArray.from_host(np.ascontiguousarray(pixel_uncertainty, np.float32))
    d_c = DevArray.from_host(np.ascontiguousarray(pixel_class, np.int32))
    d_o = DevArray.from_host(np.ascontiguousarray(offsets, np.int32)); d_p = DevArray.from_host(np.ascontiguousarray(points, np.int32))
    d_r = DevArray((S,), np.float64); d_d = DevArray((S,), np.int32); d_n = DevArray((S,), np.int32)
    _lib.check(_lib.lib().ssdr_region_stats_dev(d_u.ptr, d_c.ptr, d_o.ptr, d_p.ptr, S, class_num, _mode(sampler_args, _REG),
                                                d_r.ptr, d_d.ptr, d_n.ptr, None))
    _lib.sync()
    return d_r.to_host(), d_d.to_host(), d_n.to_host()


def dominant_labels(labels, offsets, points, num_labels):
    """ssdr_max_dominant / oracle_labeling "dominant" (sampler2.py:102-106, :127-144): label and purity per superpoint."""
    S = len(offsets) - 1
    d_l = DevArray.from_host(np.ascontiguousarray(labels, np.int32))
    d_o = DevArray.from_host(np.ascontiguousarray(offsets, np.int32)); d_p = DevArray.from_host(np.ascontiguousarray(points, np.int32))
    d_lab = DevArray((S,), np.int32); d_pur = DevArray((S,), np.float64)
    _lib.check(_lib.lib().ssdr_dominant_label_dev(d_l.ptr, d_o.ptr, d_p.ptr, S, num_labels, d_lab.ptr, d_pur.ptr, None))
    _lib.sync()
    return d_lab.to_host(), d_pur.to_host()


def add_clsbal(class_num, region_class, region_uncertainty, total_obj, skip=None):
    """sampler2.py:262-266.  The reference passes the UNLABELLED regions only (prediction(), :612-627); a caller that keeps every region in
    one array marks the others in `skip` (non-zero: not part of the population)."""
    rc = np.ascontiguousarray(region_class, np.int32)
    sel = np.ascontiguousarray(total_obj.get("selected_class_list", []), np.int32)
    d_rc = DevArray.from_host(rc); d_sel = DevArray.from_host(sel) if len(sel) else None
    d_skip = None if skip is None else DevArray.from_host(np.ascontiguousarray(skip).astype(np.uint8))
    d_u = DevArray.from_host(np.ascontiguousarray(region_uncertainty, np.float64))
    _lib.check(_lib.lib().ssdr_clsbal_dev(d_rc.ptr, len(rc), d_skip.ptr if d_skip else None, d_sel.ptr if d_sel else None, len(sel), d_u.ptr, None))
    _lib.sync()
    return d_u.to_host()


def add_classbal(class_num, region_class, region_uncertainty, skip=None):
    """sampler2.py:256-260 (`--classbal 1`): add_clsbal without the already-selected list."""
    return add_clsbal(class_num, region_class, region_uncertainty, {"selected_class_list": []}, skip)


def labeled_class_weights(dominant_label_list, class_num):
    """The draw probabilities of get_labeled_selection_cloudname_spidx_pointidx (sampler2.py:294-295): weights_percentage of the labelled
    regions' ground-truth dominant labels, normalised.  Host arithmetic on a few thousand integers (it feeds np.random.choice)."""
    lab = np.asarray(dominant_label_list, np.int64)
    dist = np.zeros([class_num])
    for c in lab:                                    # weights_percentage :92-100 as written (float64 counts)
        dist[c] = dist[c] + 1
    dist = dist / len(lab)
    w = dist[lab]
    return w / np.sum(w)


def get_labeled_selection(dominant_label_list, class_num, round_num, random_state=np.random):
    """The class-balanced draw of sampler2.py:294-302 over the labelled regions (given by their ground-truth dominant labels, in the
    reference's order: cloud by cloud as prediction() met them, ascending superpoint id): `(round_num - 1) * 1000` of them at most,
    without replacement, with NumPy's legacy generator exactly as the reference draws.  Returns the drawn positions in draw order."""
    n = len(dominant_label_list)
    batch = min((int(round_num) - 1) * 1000, n)
    if n == 0 or batch <= 0:
        return np.zeros(0, np.int64)
    return np.asarray(random_state.choice(a=n, size=batch, replace=False, p=labeled_class_weights(dominant_label_list, class_num)), np.int64)


def rank_regions(region_uncertainty):
    """sorted_inds = np.argsort(-region_uncertainty) (sampler2.py:640); equal values by ascending index."""
    u = np.ascontiguousarray(region_uncertainty, np.float64)
    d_u = DevArray.from_host(u); d_s = DevArray((len(u),), np.int32)
    _lib.check(_lib.lib().ssdr_rank_regions_dev(d_u.ptr, len(u), d_s.ptr, None))
    _lib.sync()
    return d_s.to_host()


def segment_mean_features(last_second_features, pixel_class, dom, offsets, points, sel=None):
    """compute_features (sampler2.py:333,339): mean feature of the dominant-class members of each superpoint."""
    f = np.ascontiguousarray(last_second_features, np.float32)
    n_sel = len(offsets) - 1 if sel is None else len(sel)
    d_f = DevArray.from_host(f); d_c = DevArray.from_host(np.ascontiguousarray(pixel_class, np.int32))
    d_d = DevArray.from_host(np.ascontiguousarray(dom, np.int32))
    d_o = DevArray.from_host(np.ascontiguousarray(offsets, np.int32)); d_p = DevArray.from_host(np.ascontiguousarray(points, np.int32))
    d_s = None if sel is None else DevArray.from_host(np.ascontiguousarray(sel, np.int32))
    d_out = DevArray((n_sel, f.shape[1]), np.float32)
    _lib.check(_lib.lib().ssdr_segment_mean_features_dev(d_f.ptr, f.shape[1], d_c.ptr, d_d.ptr, d_o.ptr, d_p.ptr,
                                                         d_s.ptr if d_s else None, n_sel, d_out.ptr, None))
    _lib.sync()
    return d_out.to_host()


def set_chamfer_mode(mode):
    """arithmetic of the graph's chamfer term, process-wide: "f64" (S3DIS: fps_gcn_cpu.create_cd, the default) or "f32_cuda" (the Semantic3D code's
    create_cd_cuda: float32 CUDA-kernel values, SSRD_AL_semantic3d/fps_gcn_cuda.py:13-30)"""
    _lib.check(_lib.lib().ssdr_select_set_chamfer_mode({"f64": 0, "f32_cuda": 1}[mode]))


def cloud_graph(xyz, offsets, points, sel, gcn_top=0):
    """One cloud's block of fps_adj_all (fps_gcn_cpu.py:40-117): returns (centres [n,3], cd [n,n], adj [n,n])."""
    xyz = np.ascontiguousarray(xyz, np.float32); sel = np.ascontiguousarray(sel, np.int32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    n = len(sel)
    max_sp = int((offsets[sel + 1] - offsets[sel]).max()) if n else 1
    d_x = DevArray.from_host(xyz); d_o = DevArray.from_host(offsets); d_p = DevArray.from_host(np.ascontiguousarray(points, np.int32))
    d_s = DevArray.from_host(sel)
    d_c = DevArray((n, 3), np.float64); d_dir = DevArray((n, n), np.float64); d_a = DevArray((n, n), np.float64)
    _lib.check(_lib.lib().ssdr_cloud_graph_dev(d_x.ptr, d_o.ptr, d_p.ptr, d_s.ptr, n, max(max_sp, 1), int(gcn_top), d_c.ptr, d_dir.ptr, d_a.ptr, None))
    _lib.sync()
    dirm = d_dir.to_host()
    return d_c.to_host(), dirm + dirm.T, d_a.to_host()


def create_cd(xyz, offsets, points, sel):
    """create_cd (fps_gcn_cpu.py:26-38) for the superpoints `sel` of one cloud (centred on their bbox centres)."""
    return cloud_graph(xyz, offsets, points, sel)[1]


def farthest_features_sample(feature_list, sample_number, start):
    """fps_gcn_cpu.py:119-147; `start` replaces the np.random.randint draw of :133."""
    f = np.ascontiguousarray(feature_list, np.float64)
    d_f = DevArray.from_host(f); d_o = DevArray((sample_number,), np.int32)
    _lib.check(_lib.lib().ssdr_fps_dev(d_f.ptr, f.shape[0], f.shape[1], int(start), sample_number, d_o.ptr, None))
    _lib.check(_lib.lib().ssdr_select_status(None, None))      # (a cooperative launch that was not co-resident is an error, not a result)
    _lib.sync()
    return d_o.to_host()


def farthest_superpoint_sample(xyz, offsets, points, sel, sample_number, trigger_idx):
    """sampler2.py:49-80 (the "edcd" branch) for the superpoints `sel` of one cloud (CSR instead of point lists)."""
    xyz = np.ascontiguousarray(xyz, np.float32); sel = np.ascontiguousarray(sel, np.int32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    n = len(sel)
    d_x = DevArray.from_host(xyz); d_o = DevArray.from_host(offsets); d_p = DevArray.from_host(np.ascontiguousarray(points, np.int32))
    d_s = DevArray.from_host(sel)
    d_c = DevArray((n, 3), np.float64); d_dir = DevArray((n, n), np.float64); d_a = DevArray((n, n), np.float64)
    d_out = DevArray((sample_number,), np.int32)
    L = _lib.lib()
    _lib.check(L.ssdr_cloud_graph_dev(d_x.ptr, d_o.ptr, d_p.ptr, d_s.ptr, n, int((offsets[sel + 1] - offsets[sel]).max()), 0, d_c.ptr, d_dir.ptr, d_a.ptr, None))
    _lib.check(L.ssdr_fps_superpoint_dev(d_c.ptr, d_dir.ptr, n, int(trigger_idx), sample_number, d_out.ptr, None))
    _lib.sync()
    return d_out.to_host()


def create_adj(featuresV, labeled_select_ref, unlabeled_candidate_ref, clouds):
    """gcn.create_adj (gcn.py:116-191) with the on-disk inputs replaced by `clouds` {cloud_name: (xyz [n,3] f32, offsets, points)}; refs are
    lists of {"cloud_name", "sp_idx"} as in the reference, featuresV = concatenate(unlabelled candidates, labelled regions) [N,F] (gcn.py:199).
    Returns (normalised features [N,F] float32, adjacency [N,N] float32) like the reference (which also returns its run time)."""
    f = np.ascontiguousarray(featuresV, np.float32)
    N, F = f.shape
    total_cloud, order = {}, []                 # gcn.py:122-138
    for i, r in enumerate(list(unlabeled_candidate_ref) + list(labeled_select_ref)):
        if r["cloud_name"] not in total_cloud:
            total_cloud[r["cloud_name"]] = []
            order.append(r["cloud_name"])
        total_cloud[r["cloud_name"]].append((r["sp_idx"], i))
    L = _lib.lib()
    # every cloud's centres and directed chamfer means by the batched graph call: one concatenated point array, CSR offsets shifted per cloud
    xyzs, offs, ptss, sel, rows, counts = [], [np.zeros(1, np.int32)], [], [], [], []
    pbase = sbase = 0
    for name in order:
        xyz, off, pts = clouds[name]
        off = np.ascontiguousarray(off, np.int32)
        xyzs.append(np.ascontiguousarray(xyz, np.float32)); ptss.append(np.ascontiguousarray(pts, np.int32) + pbase); offs.append(off[1:] + offs[-1][-1])
        sel += [s + sbase for s, _ in total_cloud[name]]; rows += [i for _, i in total_cloud[name]]; counts.append(len(total_cloud[name]))
        pbase += len(xyz); sbase += len(off) - 1
    counts = np.asarray(counts, np.int64)
    coff = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32); boff = np.concatenate([[0], np.cumsum(counts * counts)]).astype(np.int64)
    d_x = DevArray.from_host(np.concatenate(xyzs)); d_o = DevArray.from_host(np.concatenate(offs).astype(np.int32)); d_p = DevArray.from_host(np.concatenate(ptss))
    d_s = DevArray.from_host(np.asarray(sel, np.int32)); d_r = DevArray.from_host(np.asarray(rows, np.int32))
    d_coff = DevArray.from_host(coff); d_boff = DevArray.from_host(boff)
    d_c = DevArray((N, 3), np.float64); d_dir = DevArray((int(boff[-1]),), np.float64); d_a = DevArray((int(boff[-1]),), np.float64)
    _lib.check(L.ssdr_cloud_graph_batch_dev(d_x.ptr, d_o.ptr, d_p.ptr, d_s.ptr, d_coff.ptr, d_boff.ptr, len(order), N, int(counts.max()), 0, d_c.ptr, d_dir.ptr, d_a.ptr, None))
    d_f = DevArray.from_host(f); d_v = DevArray((N, F), np.float32); d_adj = DevArray((N, N), np.float32)
    _lib.check(L.ssdr_create_adj_dev(d_f.ptr, N, F, d_c.ptr, d_dir.ptr, d_coff.ptr, d_boff.ptr, len(order), int(counts.max()), d_r.ptr, d_v.ptr, d_adj.ptr, None))
    _lib.sync()
    return d_v.to_host(), d_adj.to_host()


class kCenterGreedy:
    """kcenterGreedy.py:46-128 (the part the AL loop calls: select_batch_ with a non-empty already_selected)."""

    def __init__(self, X, metric="euclidean"):
        assert metric == "euclidean"
        self.features = np.ascontiguousarray(X, np.float64).reshape(len(X), -1)

    def select_batch_(self, already_selected, N, **kwargs):
        a = np.ascontiguousarray(already_selected, np.int32)
        d_f = DevArray.from_host(self.features); d_a = DevArray.from_host(a); d_o = DevArray((N,), np.int32)
        _lib.check(_lib.lib().ssdr_kcenter_dev(d_f.ptr, self.features.shape[0], self.features.shape[1], d_a.ptr, len(a), N, d_o.ptr, None))
        _lib.check(_lib.lib().ssdr_select_status(None, None))
        _lib.sync()
        return list(d_o.to_host())


def GCN_FPS_sampling(labeled_select_features, labeled_select_ref, unlabeled_candidate_features, unlabeled_candidate_ref,
                     clouds, sampling_batch, gcn_number, gcn_top, start):
    """fps_gcn_cpu.py:150-178 with the on-disk inputs replaced by `clouds`:
    {cloud_name: (xyz [n,3] f32, offsets, points)}.  refs are lists of {"cloud_name", "sp_idx"} as in the reference.
    Returns {cloud_name: [sp_idx, ...]} in selection order."""
    n_unl, n_lab = len(unlabeled_candidate_ref), len(labeled_select_ref)
    N, D = n_unl + n_lab, np.asarray(unlabeled_candidate_features).shape[1]
    if int(gcn_top) > N:       # the reference's mask[source_idx, arg_idx] = 1.0 cannot broadcast (N, gcn_top) against (N, N) (:155-159)
        raise IndexError("shape mismatch: indexing arrays could not be broadcast together with shapes (%d,%d) (%d,%d)" % (N, int(gcn_top), N, N))
    V = np.concatenate([np.asarray(unlabeled_candidate_features, np.float64).reshape(n_unl, D),
                        np.asarray(labeled_select_features, np.float64).reshape(n_lab, D)])
    total_cloud, order = {}, []                 # fps_adj_all :47-62
    for i, r in enumerate(list(unlabeled_candidate_ref) + list(labeled_select_ref)):
        if r["cloud_name"] not in total_cloud:
            total_cloud[r["cloud_name"]] = []
            order.append(r["cloud_name"])
        total_cloud[r["cloud_name"]].append((r["sp_idx"], i))
    d_v = DevArray.from_host(V); d_comb = DevArray.from_host(V)
    d_tmp = [DevArray((N, D), np.float64), DevArray((N, D), np.float64)]
    blocks = []
    L = _lib.lib()
    for name in order:
        xyz, off, pts = clouds[name]
        off = np.ascontiguousarray(off, np.int32)
        sel = np.array([s for s, _ in total_cloud[name]], np.int32)
        rows = np.array([i for _, i in total_cloud[name]], np.int32)
        n = len(sel)
        d = dict(x=DevArray.from_host(np.ascontiguousarray(xyz, np.float32)), o=DevArray.from_host(off),
                 p=DevArray.from_host(np.ascontiguousarray(pts, np.int32)), s=DevArray.from_host(sel), r=DevArray.from_host(rows),
                 c=DevArray((n, 3), np.float64), d=DevArray((n, n), np.float64), a=DevArray((n, n), np.float64), n=n)
        _lib.check(L.ssdr_cloud_graph_dev(d["x"].ptr, d["o"].ptr, d["p"].ptr, d["s"].ptr, n, int((off[sel + 1] - off[sel]).max()), int(gcn_top),
                                          d["c"].ptr, d["d"].ptr, d["a"].ptr, None))
        blocks.append(d)
    src = d_v
    for hop in range(int(gcn_number)):          # :162-167
        dst = d_tmp[hop & 1]
        for d in blocks:
            _lib.check(L.ssdr_propagate_dev(d["a"].ptr, d["n"], d["r"].ptr, src.ptr, D, dst.ptr, d_comb.ptr, None))
        src = dst
    d_out = DevArray((sampling_batch,), np.int32)
    _lib.check(L.ssdr_fps_dev(d_comb.ptr, n_unl, D, int(start), sampling_batch, d_out.ptr, None))   # FPS over comb[:n_unl] (:169-170)
    _lib.check(L.ssdr_select_status(None, None))
    _lib.sync()
    file_list = {}
    for i in d_out.to_host():
        r = unlabeled_candidate_ref[int(i)]
        file_list.setdefault(r["cloud_name"], []).append(r["sp_idx"])
    return file_list
