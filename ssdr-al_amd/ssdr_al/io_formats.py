"""On-disk formats on either side of the hot path, so the accelerated path can be pointed at
the reference's directory layout:

  * binary PLY point clouds as `helper_ply.py:92-305` writes / reads them (`{cloud}.ply` under `original_ply/` and
    `input_0.040/`: x y z red green blue class),
  * the pickles of `partition/compute_superpoint.py:66-87`: `{cloud}.superpoint` = {"components", "in_component"},
    `{cloud}.gt` = float32 [2, n] pseudo labels, `total.pkl` = bookkeeping dict (`sampler2.py:194-216` reads them back),
  * the `proj_idx` pickles of `utils/data_prepare_s3dis.py:69-72` ([proj_idx, labels]).

Host-side Python like the reference's own I/O; nothing here touches the GPU.  `load_cloud` returns exactly the arrays
`pipeline.HotPath.load_rooms` / `sampler.csr_from_components` take.
"""
import os
import pickle
import sys

import numpy as np

_SCALARS = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
            "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_ORDER = {"binary_little_endian": "<", "binary_big_endian": ">"}


class PlyHeader:
    """elements in file order: [(name, count, [(property name, numpy type string) | ("list", count type, item type, name)])]"""

    def __init__(self, order, elements, data_offset):
        self.order, self.elements, self.data_offset = order, elements, data_offset

    def element(self, name):
        for e in self.elements:
            if e[0] == name:
                return e
        return None


def parse_ply_header(f):
    if f.readline().strip() != b"ply":
        raise ValueError("The file does not start whith the word ply")       # helper_ply.py:124 (message kept, typo included)
    order, elements = None, []
    while True:
        raw = f.readline()
        if raw == b"":
            raise ValueError("PLY header without end_header")
        tok = raw.split()
        if not tok or tok[0] == b"comment" or tok[0] == b"obj_info":
            continue
        if tok[0] == b"format":
            fmt = tok[1].decode()
            if fmt == "ascii":
                raise ValueError("The file is not binary")                   # helper_ply.py:129
            if fmt not in _ORDER:
                raise ValueError("unknown PLY format " + fmt)
            order = _ORDER[fmt]
        elif tok[0] == b"element":
            elements.append((tok[1].decode(), int(tok[2]), []))
        elif tok[0] == b"property":
            if not elements:
                raise ValueError("PLY property before any element")
            if tok[1] == b"list":
                elements[-1][2].append(("list", _SCALARS[tok[2].decode()], _SCALARS[tok[3].decode()], tok[4].decode()))
            else:
                elements[-1][2].append((tok[2].decode(), _SCALARS[tok[1].decode()]))
        elif tok[0] == b"end_header":
            break
    if order is None:
        raise ValueError("PLY header without a format line")
    return PlyHeader(order, elements, f.tell())


def read_ply(filename, triangular_mesh=False):
    """Structured array of the vertex element (fields named like the PLY properties); with triangular_mesh=True a list
    [vertices, faces int32[F,3]] (`helper_ply.read_ply`)."""
    with open(filename, "rb") as f:
        h = parse_ply_header(f)
        vert = h.element("vertex") or (h.elements[0] if h.elements else None)
        if vert is None:
            raise ValueError("PLY file without elements")
        if any(p[0] == "list" for p in vert[2]):
            raise ValueError("list properties on the vertex element are not supported")
        vdata = np.fromfile(f, dtype=[(n, h.order + t) for n, t in vert[2]], count=vert[1])
        if not triangular_mesh:
            return vdata
        face = h.element("face")
        if face is None:
            return [vdata, np.zeros((0, 3), np.int32)]
        (_, ct, it, _), = [p for p in face[2] if p[0] == "list"] or [(None, None, None, None)]
        if ct is None:
            raise ValueError("Unsupported faces property")
        fd = np.fromfile(f, dtype=[("k", h.order + ct), ("v1", h.order + it), ("v2", h.order + it), ("v3", h.order + it)], count=face[1])
        return [vdata, np.stack([fd["v1"], fd["v2"], fd["v3"]], 1)]


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """`helper_ply.write_ply`: every 1-D array / column of a 2-D array is one vertex property named by field_names, written in
    the machine's byte order with numpy's dtype names as PLY types; returns True, or False (after printing why) on bad input."""
    fields = list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]
    cols = []
    for a in fields:
        a = np.asarray(a)
        if a.ndim > 2:
            print("fields have more than 2 dimensions")
            return False
        cols += [a] if a.ndim < 2 else [a[:, j] for j in range(a.shape[1])]
    if len({c.shape[0] for c in cols}) > 1:
        print("wrong field dimensions")
        return False
    if len(cols) != len(field_names):
        print("wrong number of field names")
        return False
    if not filename.endswith(".ply"):
        filename += ".ply"
    n = cols[0].shape[0]
    head = ["ply", "format binary_%s_endian 1.0" % sys.byteorder, "element vertex %d" % n]
    head += ["property %s %s" % (c.dtype.name, name) for c, name in zip(cols, field_names)]
    if triangular_faces is not None:
        head += ["element face %d" % triangular_faces.shape[0], "property list uchar int vertex_indices"]
    head.append("end_header")
    rec = np.empty(n, dtype=[(name, c.dtype.str) for c, name in zip(cols, field_names)])
    for c, name in zip(cols, field_names):
        rec[name] = c
    with open(filename, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        rec.tofile(f)
        if triangular_faces is not None:
            tf = np.asarray(triangular_faces).astype(np.int32)
            fr = np.empty(tf.shape[0], dtype=[("k", "u1"), ("0", "i4"), ("1", "i4"), ("2", "i4")])
            fr["k"] = 3
            fr["0"], fr["1"], fr["2"] = tf[:, 0], tf[:, 1], tf[:, 2]
            fr.tofile(f)
    return True


# ---- pickles of the partition / sampling stages ----------------------------------------------------------------------
def save_superpoint(path, components, in_component):
    """compute_superpoint.py:66-70: components = object array of point-index lists, in_component = component id per point"""
    comp = np.empty(len(components), dtype=object)
    for i, c in enumerate(components):
        comp[i] = list(int(x) for x in c)
    with open(path, "wb") as f:
        pickle.dump({"components": comp, "in_component": np.asarray(in_component)}, f)


def load_superpoint(path):
    with open(path, "rb") as f:
        sp = pickle.load(f)
    return sp["components"], np.asarray(sp["in_component"])


def save_gt(path, pseudo_gt):
    """{cloud}.gt: float32 [2, n] — row 0 labelled flag, row 1 pseudo label (compute_superpoint.py:72-74, sampler2.py:127-144)"""
    with open(path, "wb") as f:
        pickle.dump(np.asarray(pseudo_gt, np.float32), f)


def load_gt(path):
    with open(path, "rb") as f:
        return np.asarray(pickle.load(f))


def new_total(cloud_components):
    """total.pkl as compute_superpoint.py writes it for {cloud: components}: unlabeled superpoint ids per cloud + counters"""
    total = {"unlabeled": {}, "file_num": 0, "sp_num": 0, "point_num": 0}
    for name, comp in cloud_components.items():
        total["unlabeled"][name] = np.arange(len(comp))
        total["file_num"] += 1
        total["sp_num"] += len(comp)
        total["point_num"] += int(sum(len(c) for c in comp))
    return total


def save_total(path, total):
    with open(path, "wb") as f:
        pickle.dump(total, f)


def load_total(path):
    with open(path, "rb") as f:
        return pickle.load(f)


def save_proj(path, proj_idx, labels):
    """{cloud}_proj.pkl (data_prepare_s3dis.py:69-72): nearest sub-sampled point of every raw point + the raw labels"""
    with open(path, "wb") as f:
        pickle.dump([np.asarray(proj_idx), np.asarray(labels)], f)


def load_proj(path):
    with open(path, "rb") as f:
        proj_idx, labels = pickle.load(f)
    return np.asarray(proj_idx), np.asarray(labels)


def load_cloud(data_path, cloud_name, sub_dir="input_0.040", superpoint_dir="superpoint"):
    """One cloud of the reference's layout: <data_path>/<sub_dir>/<cloud>.ply (+ <data_path>/<superpoint_dir>/<cloud>.superpoint when
    present) -> (xyz f32[n,3], rgb u8[n,3], labels i32[n], components or None)."""
    d = read_ply(os.path.join(data_path, sub_dir, cloud_name + ".ply"))
    xyz = np.stack([d["x"], d["y"], d["z"]], 1).astype(np.float32)
    rgb = np.stack([d["red"], d["green"], d["blue"]], 1)
    labels = np.asarray(d["class"]).astype(np.int32)
    sp_path = os.path.join(data_path, superpoint_dir, cloud_name + ".superpoint")
    comps = load_superpoint(sp_path)[0] if os.path.exists(sp_path) else None
    return xyz, rgb, labels, comps
