"""TEST INFRASTRUCTURE ONLY — NumPy restatement of the Semantic3D sampling loader's partition of a cloud into network inputs
(/root/reference/SSRD_AL_semantic3d/semantic3d_dataset_sampling.py: split3 :198-236, the merge rule of tf_map :243-255).
Pinned by tests/golden/split3_golden.npz, made by calling the reference's own method (tests/golden/make_golden_split3.py).

The reference collects a part as `list(x & y & z)` over Python sets of ints: the order of the indices inside a part is the iteration
order of CPython's set (hash-table layout, interpreter-version dependent).  Here — and in the product — a part lists its points by
ascending index, sub-parts in the order split3 appends them; the golden vectors compare parts as index SETS, in part order."""
import numpy as np


def split3(batch_xyz, source_idx, part_list, max_size=800000, recurse_max_size=800000):
    """:198-236.  recurse_max_size: the reference's recursive call passes the literal 800000 whatever the caller's max_size was."""
    x_min = float(np.min(batch_xyz[:, 0])); x_max = float(np.max(batch_xyz[:, 0])); x_len = x_max - x_min
    y_min = float(np.min(batch_xyz[:, 1])); y_max = float(np.max(batch_xyz[:, 1])); y_len = y_max - y_min
    z_min = float(np.min(batch_xyz[:, 2])); z_max = float(np.max(batch_xyz[:, 2])); z_len = z_max - z_min
    x1 = batch_xyz[:, 0] < x_min + 0.5 * x_len          # float32 column against a Python float: a float32 comparison
    y1 = batch_xyz[:, 1] < y_min + 0.5 * y_len
    z1 = batch_xyz[:, 2] < z_max + 0.5 * z_len          # :224 as written (z_max, not z_min): true for every point
    for x in (x1, ~x1):
        for y in (y1, ~y1):
            for z in (z1, ~z1):
                cur = np.flatnonzero(x & y & z)
                part = source_idx[cur]
                if len(cur) > max_size:
                    split3(batch_xyz[cur], part, part_list, recurse_max_size, recurse_max_size)
                else:
                    part_list.append(part)


def combine(part_list, merge_max=2000):
    """tf_map :243-255: a part of at most merge_max points joins the one before it; empty results are dropped."""
    out = []
    for part in part_list:
        if len(part) > merge_max:
            out.append(part)
        elif len(out) > 0:
            out[-1] = np.concatenate([out[-1], part], axis=0)
        else:
            out.append(part)
    return [p for p in out if len(p) > 0]


def parts(xyz, max_size=800000, merge_max=2000):
    pl = []
    split3(np.asarray(xyz, np.float32), np.arange(len(xyz)), pl, max_size, max_size)
    return combine(pl, merge_max)
