"""On-disk formats: files written by the reference's own code (tests/golden/formats, made by
tests/golden/make_golden_formats.py) are read back exactly, and our writers reproduce them byte for byte."""
import os
import pickle

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden", "formats")


@pytest.fixture(scope="module")
def exp():
    return np.load(os.path.join(G, "expected.npz"))


def test_read_reference_ply(exp):
    from ssdr_al import io_formats as io
    d = io.read_ply(os.path.join(G, "cloud.ply"))
    assert d.dtype.names == ("x", "y", "z", "red", "green", "blue", "class")
    assert np.array_equal(np.stack([d["x"], d["y"], d["z"]], 1), exp["xyz"]) and d["x"].dtype == np.float32
    assert np.array_equal(np.stack([d["red"], d["green"], d["blue"]], 1), exp["rgb"]) and d["red"].dtype == np.uint8
    assert np.array_equal(d["class"], exp["lab"])
    v, f = io.read_ply(os.path.join(G, "mesh.ply"), triangular_mesh=True)
    assert v["x"].dtype == np.float64 and np.array_equal(v["scalar"], exp["lab"].astype(np.int32))
    assert np.array_equal(f, exp["faces"])


def test_write_ply_is_byte_identical_to_the_reference_writer(tmp_path, exp):
    from ssdr_al import io_formats as io
    p = str(tmp_path / "cloud")                                   # extension appended like the reference does
    assert io.write_ply(p, [exp["xyz"], exp["rgb"], exp["lab"]], ["x", "y", "z", "red", "green", "blue", "class"]) is True
    assert open(p + ".ply", "rb").read() == open(os.path.join(G, "cloud.ply"), "rb").read()
    m = str(tmp_path / "mesh.ply")
    assert io.write_ply(m, [exp["xyz"].astype(np.float64), exp["lab"].astype(np.int32)], ["x", "y", "z", "scalar"], triangular_faces=exp["faces"])
    assert open(m, "rb").read() == open(os.path.join(G, "mesh.ply"), "rb").read()
    # the reference's refusals: message printed, False returned
    assert io.write_ply(p, [exp["xyz"], exp["lab"][:-1]], ["x", "y", "z", "c"]) is False
    assert io.write_ply(p, [exp["xyz"]], ["x", "y"]) is False
    assert io.write_ply(p, [np.zeros((2, 2, 2))], ["a"]) is False


def test_ply_errors(tmp_path):
    from ssdr_al import io_formats as io
    bad = tmp_path / "a.ply"
    bad.write_bytes(b"plx\n")
    with pytest.raises(ValueError, match="does not start"):
        io.read_ply(str(bad))
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0.5\n")
    with pytest.raises(ValueError, match="not binary"):
        io.read_ply(str(bad))
    # big-endian files and comment lines are read too
    be = tmp_path / "be.ply"
    be.write_bytes(b"ply\nformat binary_big_endian 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty uchar c\nend_header\n"
                   + np.array([(1.5, 7), (-2.0, 9)], dtype=[("x", ">f4"), ("c", "u1")]).tobytes())
    d = io.read_ply(str(be))
    assert d["x"].tolist() == [1.5, -2.0] and d["c"].tolist() == [7, 9]


def test_superpoint_and_gt_pickles(tmp_path, exp):
    from ssdr_al import io_formats as io, sampler
    comps, inc = io.load_superpoint(os.path.join(G, "cloud.superpoint"))
    assert np.array_equal(inc, exp["in_component"]) and [len(c) for c in comps] == exp["comp_len"].tolist()
    assert np.array_equal(np.concatenate([np.asarray(c) for c in comps]), exp["comp_flat"])
    off, pts = sampler.csr_from_components(comps)                  # what the GPU path consumes
    assert off.tolist() == np.concatenate([[0], np.cumsum(exp["comp_len"])]).tolist() and np.array_equal(pts, exp["comp_flat"])
    # our writers produce the same pickled objects as the reference's code did
    io.save_superpoint(str(tmp_path / "c.superpoint"), comps, inc)
    a = pickle.load(open(tmp_path / "c.superpoint", "rb")); b = pickle.load(open(os.path.join(G, "cloud.superpoint"), "rb"))
    assert a.keys() == b.keys() and a["components"].dtype == object and [list(x) for x in a["components"]] == [list(x) for x in b["components"]]
    assert np.array_equal(a["in_component"], b["in_component"]) and a["in_component"].dtype == b["in_component"].dtype
    gt = io.load_gt(os.path.join(G, "cloud.gt"))
    assert gt.shape == (2, len(exp["lab"])) and gt.dtype == np.float32 and not gt.any()
    io.save_gt(str(tmp_path / "c.gt"), gt)
    assert open(tmp_path / "c.gt", "rb").read() == open(os.path.join(G, "cloud.gt"), "rb").read()
    total = io.new_total({"cloud": comps})
    assert total["file_num"] == 1 and total["sp_num"] == 5 and total["point_num"] == len(exp["lab"]) and total["unlabeled"]["cloud"].tolist() == [0, 1, 2, 3, 4]
    io.save_total(str(tmp_path / "total.pkl"), total)
    assert io.load_total(str(tmp_path / "total.pkl"))["sp_num"] == 5
    io.save_proj(str(tmp_path / "c_proj.pkl"), np.arange(4, dtype=np.int32), np.ones(4, np.uint8))
    pi, lb = io.load_proj(str(tmp_path / "c_proj.pkl"))
    assert pi.tolist() == [0, 1, 2, 3] and lb.tolist() == [1, 1, 1, 1]


def test_load_cloud_from_the_reference_layout(tmp_path, exp):
    from ssdr_al import io_formats as io
    os.makedirs(tmp_path / "input_0.040"); os.makedirs(tmp_path / "superpoint")
    for src, dst in (("cloud.ply", "input_0.040/Area_5_office_1.ply"), ("cloud.superpoint", "superpoint/Area_5_office_1.superpoint")):
        (tmp_path / dst).write_bytes(open(os.path.join(G, src), "rb").read())
    xyz, rgb, lab, comps = io.load_cloud(str(tmp_path), "Area_5_office_1")
    assert np.array_equal(xyz, exp["xyz"]) and np.array_equal(rgb, exp["rgb"]) and np.array_equal(lab, exp["lab"]) and lab.dtype == np.int32
    assert len(comps) == 5
