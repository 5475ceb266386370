"""Generates the golden vectors under tests/golden/ from the REAL reference ops.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
It drives the reference's own C++ (grid_subsampling.cpp, knn_.cxx + nanoflann.hpp), compiled from where it
lies by oracle/Makefile into oracle/_ref/libssdr_ref.so, on seeded inputs and stores inputs + expected outputs.
The .npz files are data only; nothing of the reference's source travels.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def knn_cases(rng):
    n = 1536
    uni = rng.random((n, 3), dtype=np.float32)
    room = rng.random((n, 3), dtype=np.float32) * np.array([6, 4, 3], np.float32)
    room[: n // 2, 2] = 0.0                     # floor
    room[n // 2:, 0] = 6.0                      # wall
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(10), indexing="ij"), -1).reshape(-1, 3)
    lattice = (g.astype(np.float32) * np.float32(0.04))[rng.permutation(len(g))]
    base = rng.random((1000, 3), dtype=np.float32)
    dup = np.concatenate([base, base[rng.integers(0, 1000, n - 1000)]])[rng.permutation(n)]   # padded tile
    return {"uniform": uni, "room": room, "lattice": lattice, "duplicate_padded": dup,
            "tiny7": rng.random((7, 3), dtype=np.float32),
            "all_same": np.tile(rng.random((1, 3), dtype=np.float32), (64, 1))}


def main():
    ref = oracle.ref()
    assert ref is not None, "reference build missing (needs /root/reference)"
    rng = np.random.default_rng(20241008)
    out = {}
    for name, p in knn_cases(rng).items():
        sub = p[: max(1, len(p) // 4)]
        out[name + "/pts"] = p
        out[name + "/self16"] = ref.knn(p, p, 16).astype(np.int32)
        out[name + "/self1"] = ref.knn(p, p, 1).astype(np.int32)
        out[name + "/up1"] = ref.knn(sub, p, 1).astype(np.int32)          # tf_map's interp query
        out[name + "/self5"] = ref.knn(p, p, 5).astype(np.int32)
    b = rng.random((3, 700, 3), dtype=np.float32)
    q = (rng.random((3, 150, 3), dtype=np.float32) - 0.25) * 2            # some queries outside the support box
    out["batch/pts"], out["batch/q"] = b, q
    out["batch/idx16"] = ref.knn_batch(b, q, 16, omp=True).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "knn_golden.npz"), **out)

    # full 5-level pyramid exactly as tf_map builds it (s3dis_dataset.py:164-177), N=2048
    xyz = knn_cases(rng)["room"][None, :1536]
    xyz = np.concatenate([xyz, rng.random((1, 1536, 3), dtype=np.float32)], 0)
    pyr = {"xyz": xyz, "ratios": np.array([4, 4, 4, 4, 2], np.int32)}
    cur = xyz
    for i, r in enumerate([4, 4, 4, 4, 2]):
        neigh = ref.knn_batch(cur, cur, 16, omp=True).astype(np.int32)
        sub = cur[:, : cur.shape[1] // r]
        pyr["neigh%d" % i] = neigh
        pyr["sub%d" % i] = neigh[:, : cur.shape[1] // r]
        pyr["interp%d" % i] = ref.knn_batch(sub, cur, 1, omp=True).astype(np.int32)
        cur = sub
    np.savez_compressed(os.path.join(HERE, "pyramid_golden.npz"), **pyr)

    # grid subsampling
    g = {}
    # (1) the 13-point hand case of SURVEY appendix A.4 (distinct voxels along x, dl = 1) + a label tie
    x = np.array([0.5, 1.5, 2.55, 3.5, 5.5, 7.5, 9.5, 11.5, 13.5, 15.5, 20.5, 30.5, 2.45], np.float32)
    hand = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)
    g["hand/pts"] = hand
    g["hand/out"] = ref.grid_subsampling(hand, None, None, 1.0)[0]
    tie = np.tile(np.array([[0.2, 0.2, 0.2]], np.float32), (8, 1))
    for nm, labs in (("tieA", [3, 7, 3, 7, 1, 1, 9, 9]), ("tieB", [7, 3, 7, 3, 9, 9, 1, 1])):
        lab = np.array(labs, np.int32)
        g[nm + "/pts"], g[nm + "/cls"] = tie, lab
        g[nm + "/out_cls"] = ref.grid_subsampling(tie, None, lab, 1.0)[1]
    # (2) synthetic room, u8 colours as float + labels, dl = 0.04 (data_prepare_s3dis.py:58)
    n = 6000
    pts = (rng.random((n, 3), dtype=np.float32) * np.array([3, 2, 1.5], np.float32) - np.array([1, 0.5, 0.2], np.float32)).astype(np.float32)
    pts[: n // 2, 2] = np.float32(-0.2) + rng.normal(0, 0.002, n // 2).astype(np.float32)
    col = rng.integers(0, 256, (n, 3)).astype(np.float32)
    lab = rng.integers(0, 13, n).astype(np.int32)
    rp, rf, rc = ref.grid_subsampling(pts, col, lab, 0.04)
    g["room/pts"], g["room/col"], g["room/lab"] = pts, col, lab
    g["room/out_pts"], g["room/out_col"], g["room/out_lab"] = rp, rf, rc
    # (3) degenerate: one voxel; negative coordinates; a point exactly on a voxel face; > 13 labels, 2 columns
    one = np.tile(np.array([[0.1, 0.2, 0.3]], np.float32), (300, 1))
    g["one/pts"], g["one/out"] = one, ref.grid_subsampling(one, None, None, 0.1)[0]
    neg = (-rng.random((1500, 3), dtype=np.float32) * 5).astype(np.float32)
    neg[:10] = np.round(neg[:10] / np.float32(0.3)) * np.float32(0.3)    # on faces
    g["neg/pts"], g["neg/out"] = neg, ref.grid_subsampling(neg, None, None, 0.3)[0]
    ml = rng.random((1200, 3), dtype=np.float32)
    ml_lab = rng.integers(-4, 24, (1200, 2)).astype(np.int32)
    g["manylab/pts"], g["manylab/cls"] = ml, ml_lab
    o = ref.grid_subsampling(ml, None, ml_lab, 0.5)
    g["manylab/out_pts"], g["manylab/out_cls"] = o
    np.savez_compressed(os.path.join(HERE, "subsample_golden.npz"), **g)
    for f in ("knn_golden.npz", "pyramid_golden.npz", "subsample_golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def distance_pick_golden():
    """knn_batch_distance_pick of the REAL reference with the clock its std::mt19937 is seeded from pinned (oracle/ref_shim.cpp)."""
    import oracle
    rng = np.random.default_rng(21)
    pts = (rng.random((3, 700, 3)) * np.array([4, 3, 2])).astype(np.float32)
    pts[1, 300:] = pts[1, :400]                               # duplicated points: equidistant neighbours
    out = {"dp/pts": pts}
    for seed, nq, K in ((1700000000, 120, 16), (7, 64, 1)):
        idx, q = oracle.ref().knn_batch_distance_pick(pts, nq, K, seed)
        out["dp/%d/idx" % seed], out["dp/%d/q" % seed] = idx, q
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "distance_pick_golden.npz"), **out)
    print("distance_pick_golden.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if os.environ.get("SSDR_GOLDEN_ONLY") != "distance_pick":       # the other files are byte-stable; regenerate them only on purpose
        main()
    distance_pick_golden()
