"""Mirror of the evaluation tail of the reference ("next" row N2): vote smoothing, re-projection, confusion, IoU
(/root/reference/SSDR_AL_s3dis/RandLANet.py:326-334, 353-411; helper_tool.py:237-262; utils/data_prepare_s3dis.py:69)."""
import numpy as np

from . import _lib
from ._lib import DevArray
from .knn import knn


def project_indices(sub_xyz, xyz):
    """proj_idx = np.squeeze(KDTree(sub_xyz).query(xyz, return_distance=False)) (data_prepare_s3dis.py:69-70): nearest
    sub-sampled point of every raw point.  Exact nearest neighbour in float32 arithmetic; where two sub-points are
    equidistant the reference's sklearn tree may pick the other one."""
    return knn(sub_xyz, xyz, 1)[:, 0].astype(np.int32)


class VoteAccumulator:
    """test_probs of one cloud, resident on the device (RandLANet.py:292-334)."""

    def __init__(self, num_points, num_classes, test_smooth=0.95):
        self.C, self.smooth = num_classes, float(test_smooth)
        self.test_probs = DevArray.from_host(np.zeros((num_points, num_classes), np.float32))
        self.owner = DevArray.from_host(np.full(num_points, -1, np.int32))

    def update(self, p_idx, probs):
        """test_probs[p_idx] = smooth * test_probs[p_idx] + (1 - smooth) * probs (:333)."""
        d_i = DevArray.from_host(np.ascontiguousarray(p_idx, np.int32)); d_p = DevArray.from_host(np.ascontiguousarray(probs, np.float32))
        _lib.check(_lib.lib().ssdr_vote_smooth_dev(self.test_probs.ptr, d_i.ptr, d_p.ptr, len(p_idx), self.C, self.smooth, self.owner.ptr, None))
        _lib.sync()

    def probs(self):
        return self.test_probs.to_host()

    def confusion(self, labels, proj_idx=None):
        """(preds, confusion int64 [C,C], IoU float64 [C]) on the sub-cloud, or on the raw cloud through proj_idx (:353-405)."""
        lab = np.ascontiguousarray(labels, np.int32)
        n = len(lab)
        d_l = DevArray.from_host(lab)
        d_proj = None if proj_idx is None else DevArray.from_host(np.ascontiguousarray(proj_idx, np.int32))
        d_pred = DevArray((n,), np.int32); d_conf = DevArray.from_host(np.zeros((self.C, self.C), np.uint64)); d_iou = DevArray((self.C,), np.float64)
        _lib.check(_lib.lib().ssdr_confusion_dev(self.test_probs.ptr, self.C, d_proj.ptr if d_proj else None, d_l.ptr, n, d_pred.ptr, d_conf.ptr, d_iou.ptr, None))
        _lib.sync()
        return d_pred.to_host(), d_conf.to_host().astype(np.int64), d_iou.to_host()


def IoU_from_confusions(confusions):
    """helper_tool.py:237-262 for one [C,C] matrix, computed by the device kernel."""
    c = np.ascontiguousarray(confusions, np.uint64)
    C = c.shape[-1]
    d_conf = DevArray.from_host(c); d_iou = DevArray((C,), np.float64)
    d_dummy = DevArray((1,), np.int32); d_p = DevArray((C,), np.float32)
    _lib.check(_lib.lib().ssdr_confusion_dev(d_p.ptr, C, None, d_dummy.ptr, 0, None, d_conf.ptr, d_iou.ptr, None))
    _lib.sync()
    return d_iou.to_host()
