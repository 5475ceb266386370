"""The sizes the reference really feeds the path outside the 40960-point training tiles: the AL round's "sampling"
mode runs B = 1 with N = every point of the (sub-sampled) room (S3/s3dis_dataset.py:129-131), Semantic3D parts hold up to 800 000 points
(SSRD_AL_semantic3d/semantic3d_dataset_sampling.py:198-253).  KNN pyramid bit-exact against the REAL reference ops where the compiled
reference travelled with the snapshot (oracle/_ref, cpp_knn_omp over the queries), else the C oracle; the network against the NumPy oracle
at 100 352 points and through size-independent properties at 800 256.  GPU only: the CPU logic build is far too slow at these sizes."""
import numpy as np
import pytest

from conftest import assert_bits_equal

RATIOS = [4, 4, 4, 4, 2]


def _room_cloud(n, seed):
    """room-like: floor / ceiling / two walls on a jittered 4 cm lattice, a few boxes, one block duplicated (the padded-tile tie path), shuffled"""
    rng = np.random.default_rng(seed)
    side = int(np.sqrt(n / 3.2)) + 1
    u, v = np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32), indexing="ij")
    uv = np.stack([u.ravel(), v.ravel()], 1) * np.float32(0.04)
    L = np.float32(side * 0.04)
    planes = [np.concatenate([uv, np.zeros((len(uv), 1), np.float32)], 1),                                   # floor
              np.concatenate([uv, np.full((len(uv), 1), 3.0, np.float32)], 1),                               # ceiling
              np.stack([uv[:, 0], np.zeros(len(uv), np.float32), uv[:, 1] * np.float32(3.0) / L], 1),        # wall y = 0
              np.stack([np.zeros(len(uv), np.float32), uv[:, 0], uv[:, 1] * np.float32(3.0) / L], 1)]        # wall x = 0
    p = np.concatenate(planes)[: n - 6000].astype(np.float32)
    p += rng.normal(0, 0.002, p.shape).astype(np.float32)
    p = np.concatenate([p, p[:6000]])                 # duplicated block: exact ties, answered by the tree hand-over
    assert len(p) == n
    return p[rng.permutation(n)]


def _checker():
    import oracle
    r = oracle.ref()
    if r is not None:
        return "reference cpp_knn_omp", lambda s, q, k: r.knn(s, q, k, omp=True)
    o = oracle.c()
    return "C oracle", lambda s, q, k: o.knn_batch(s[None], q[None], k, threads=1)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300032, 800256])
def test_pyramid_at_whole_room_and_semantic3d_part_sizes(n):
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    from ssdr_al import _lib, knn
    _lib.use(GPU_LIB)
    try:
        xyz = _room_cloud(n, 41 + n % 7)
        neigh, sub, interp = knn.knn_pyramid(xyz[None], RATIOS, 16)
        st = knn.knn_status()
        name, ref_knn = _checker()
        print("\nN = %d: rows handed to the tree walk K=16 / K=1: %d / %d, status bits %d, deepest tree %d; checker: %s" % (n, st[0], st[1], st[2], st[3], name))
        assert st[2] == 0, "a device-side capacity overflowed (status bits %d)" % st[2]
        cur = xyz
        for i, r in enumerate(RATIOS):
            assert_bits_equal(neigh[i][0], ref_knn(cur, cur, 16).astype(np.int32), "N %d level %d neigh" % (n, i))
            nxt = cur[: len(cur) // r]
            assert_bits_equal(interp[i][0], ref_knn(nxt, cur, 1).astype(np.int32), "N %d level %d interp" % (n, i))
            assert np.array_equal(sub[i][0], neigh[i][0][: len(nxt)])
            cur = nxt
    finally:
        _lib.use(None)


@pytest.mark.gpu
def test_network_at_100k_points_against_oracle():
    """one whole sub-sampled room as ONE cloud (B = 1, N = 100 352): exact-f32 and split-bf16 products inside the 1e-3 bar"""
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    import oracle
    from oracle import randla_np as R
    from ssdr_al import _lib, randlanet
    _lib.use(GPU_LIB)
    try:
        n = 100352
        xyz0 = _room_cloud(n, 5)[None]
        rng = np.random.default_rng(6)
        feat = np.concatenate([xyz0 - xyz0.mean(1, keepdims=True), rng.random((1, n, 3), dtype=np.float32)], -1)
        name, ref_knn = _checker()
        pyr = R.build_pyramid(xyz0, RATIOS, lambda s, q, k: ref_knn(s[0], q[0], k)[None])
        W = R.init_weights(0)
        p, f = R.forward(W, feat, *pyr, dtype=np.float32)
        for mode in ("f32", "bf16x3"):
            gp, gf = randlanet.Network().load(W).set_precision(mode).infer(feat, xyz0)
            ep, ef = np.abs(gp - p).max(), np.abs(gf - f).max()
            print("\nN = %d, %s: max |probs - oracle| = %.3g, max |feat32 - oracle| = %.3g (|feat| max %.3g)" % (n, mode, ep, ef, np.abs(f).max()))
            assert ep < 1e-3 and ef < 1e-3, (mode, ep, ef)
    finally:
        _lib.use(None)


@pytest.mark.gpu
def test_network_at_800k_points_properties():
    """a Semantic3D part of 800 256 points as one cloud: finite outputs, probabilities a distribution, and the two independent kernel families
    (exact-f32 16 x 16 tiles, split-bf16 32 x 32 tiles) inside 2e-3 of each other (each is inside 1e-3 of the fp32 oracle where that can run)"""
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    from oracle import randla_np as R
    from ssdr_al import _lib, randlanet
    from ssdr_al.helper_tool import ConfigSemantic3D
    _lib.use(GPU_LIB)
    try:
        n = 800256
        xyz0 = _room_cloud(n, 9)[None]
        rng = np.random.default_rng(10)
        feat = np.concatenate([xyz0 - xyz0.mean(1, keepdims=True), rng.random((1, n, 3), dtype=np.float32)], -1)
        W = R.init_weights(0, num_classes=ConfigSemantic3D.num_classes)
        out = {}
        for mode in ("f32", "bf16x3"):
            gp, gf = randlanet.Network(ConfigSemantic3D).load(W).set_precision(mode).infer(feat, xyz0)
            assert gp.shape == (n, 8) and gf.shape == (n, 32)
            assert np.isfinite(gp).all() and np.isfinite(gf).all()
            assert np.abs(gp.sum(1) - 1).max() < 1e-5 and gp.min() >= 0
            out[mode] = (gp, gf)
        dp, df = np.abs(out["f32"][0] - out["bf16x3"][0]).max(), np.abs(out["f32"][1] - out["bf16x3"][1]).max()
        print("\nN = %d: f32 vs bf16x3 kernels: max |d probs| = %.3g, max |d feat32| = %.3g" % (n, dp, df))
        assert dp < 2e-3 and df < 2e-3
    finally:
        _lib.use(None)
