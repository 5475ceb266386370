"""Worker of tests/test_sanitizers.py: a small pass of every stage through the AddressSanitizer + UBSan builds of the CPU logic
library (tests/hipemu/libssdr_al_emu_san.so) and of the C oracle (oracle/liboracle_san.so).  Runs with libasan preloaded."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))


def main():
    import oracle
    from oracle import pipeline_np
    from oracle import randla_np as R
    from ssdr_al import _lib, knn, pipeline, subsampling, synthetic
    from ssdr_al.helper_tool import ConfigS3DIS
    _lib.use(os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu_san.so"))
    o = oracle.c()
    rng = np.random.default_rng(0)
    # grid subsample (incl. reference row order) + KNN incl. tie rows (duplicates -> the tree hand-over) and generic K
    raw = (rng.random((6000, 3), dtype=np.float32) * np.array([3, 2, 1], np.float32)).astype(np.float32)
    col = rng.integers(0, 256, (6000, 3)).astype(np.float32); lab = rng.integers(0, 13, 6000).astype(np.int32)
    got = subsampling.compute(raw, features=col, classes=lab, sampleDl=0.05)
    exp = o.grid_subsampling(raw, col, lab, 0.05)
    assert all(np.array_equal(a, b) for a, b in zip(got, exp))
    p = got[0][:600].copy(); p[-100:] = p[:100]
    assert np.array_equal(knn.knn(p, p, 16), o.knn(p, p, 16))
    assert np.array_equal(knn.knn(p[:300], p, 1), o.knn(p[:300], p, 1))
    assert np.array_equal(knn.knn(p, p[:40], 5), o.knn(p, p[:40], 5))
    # the whole hot path on one tiny room (tile, pyramid, network in split-bf16, scoring, selection with a keep-top mask)

    class Cfg(ConfigS3DIS):
        num_points = 512
    W = R.init_weights(0)
    rooms = [synthetic.make_room(5200, density=25.0)]
    hp = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2, gcn_top=3, precision="bf16x3").load_rooms(rooms)
    sel, unl = hp.step()
    ref = pipeline_np.run(hp, rooms, W, threads=1, net_outputs=(hp.probs.to_host(), hp.f32.to_host()))
    assert np.array_equal(sel, ref["selected"])
    print("sanitized pass ok")


if __name__ == "__main__":
    main()
