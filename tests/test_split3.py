"""split3 + the <= 2000 merge rule of the Semantic3D sampling loader (SSRD_AL_semantic3d/semantic3d_dataset_sampling.py:198-255) as a device
partition (ssdr_split3_dev): golden vectors from the reference's own method (tests/golden/make_golden_split3.py), the NumPy oracle, and the
product on the CPU logic build and the GPU.  Parts are compared as index sets in part order (see oracle/split3_np.py for why)."""
import numpy as np
import pytest

# name: (seed, points, max_size, merge_max)
CASES = {
    "small": (3, 12000, 3000, 500),            # one quadrant above max_size (recursed once: the reference's recursive call passes 800000, so its 6933-point
                                               # sub-part stays whole), two quadrants at most merge_max points (merged)
    "first_small": (4, 9000, 800000, 2000),    # the FIRST part is small: it opens the list and the next parts are judged on their own
    "part800k": (5, 1000000, 800000, 2000),    # the real sizes: a quadrant of more than 800 000 points is split again
}


def cloud(seed, n):
    """a scan-like cloud: most points in one dense corner cluster, the rest spread thin; the z range is wide so that the reference's
    `z < z_max + 0.5 * z_len` (:224) is visibly not a split"""
    rng = np.random.default_rng(seed)
    if seed == 4:
        k = 700
        a = rng.random((k, 3)) * np.array([4.0, 4.0, 30.0])                        # few points in the low-x low-y quadrant
        b = rng.random((n - k, 3)) * np.array([40.0, 40.0, 30.0]) + np.array([30.0, 30.0, 0.0])
        return np.concatenate([a, b]).astype(np.float32)[rng.permutation(n)]
    dense = int(n * 0.86)
    a = rng.random((dense, 3)) * np.array([30.0, 25.0, 12.0])
    b = rng.random((n - dense - 300, 3)) * np.array([100.0, 80.0, 40.0])
    c = rng.random((300, 3)) * np.array([10.0, 10.0, 40.0]) + np.array([85.0, 5.0, 0.0])      # a thin far quadrant
    return np.concatenate([a, b, c]).astype(np.float32)[rng.permutation(n)]


def _check(parts, g, name):
    assert [len(p) for p in parts] == g[name + "/sizes"].tolist()
    assert [int(np.asarray(p, np.int64).sum()) for p in parts] == g[name + "/sum"].tolist()
    assert [int((np.asarray(p).astype(np.uint64) ** 2).sum() % (1 << 62)) for p in parts] == g[name + "/sumsq"].tolist()
    if name + "/sorted" in g:
        assert np.array_equal(np.concatenate([np.sort(p) for p in parts]), g[name + "/sorted"])


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_split3_against_reference(golden, name):
    from oracle import split3_np
    g = golden("split3_golden.npz")
    seed, n, max_size, merge_max = CASES[name]
    xyz = cloud(seed, n)
    raw = []
    split3_np.split3(xyz, np.arange(n), raw, max_size)
    assert [len(p) for p in raw] == g[name + "/raw_sizes"].tolist()          # the eight parts per call, the four z twins empty
    _check(split3_np.combine(raw, merge_max), g, name)


@pytest.mark.parametrize("name", list(CASES))
def test_split3_device_partition(golden, backend, name):
    from ssdr_al import semantic3d_sampling as S3
    g = golden("split3_golden.npz")
    seed, n, max_size, merge_max = CASES[name]
    if backend == "emu" and n > 200000:
        pytest.skip("the CPU logic build runs the small cases")
    xyz = cloud(seed, n)
    parts, part_of = S3.split3_parts(xyz, max_size=max_size, merge_max=merge_max, return_part_ids=True)
    _check(parts, g, name)
    for p in parts:                                   # canonical order: ascending index inside a leaf, leaves in append order
        assert len(np.unique(p)) == len(p)
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(n))
    for k, p in enumerate(parts):
        assert (part_of[p] == k).all()


def test_split3_refuses_coincident_points(backend):
    """more than max_size coincident points never split: the reference recurses without end, the device partition says so"""
    from ssdr_al import _lib, semantic3d_sampling as S3
    xyz = np.zeros((5000, 3), np.float32)
    with pytest.raises(_lib.SsdrError):
        S3.split3_parts(xyz, max_size=1000)


def test_split3_tiny_and_degenerate_clouds(backend):
    """a handful of points (every part at most merge_max: one combined part), points on a line (empty quadrants), one point"""
    from oracle import split3_np
    from ssdr_al import semantic3d_sampling as S3
    rng = np.random.default_rng(8)
    for xyz in (rng.random((5, 3)).astype(np.float32),
                np.stack([np.linspace(0, 1, 50), np.zeros(50), np.zeros(50)], 1).astype(np.float32),
                np.zeros((1, 3), np.float32),
                np.concatenate([rng.random((3000, 3)), rng.random((40, 3)) + 5.0]).astype(np.float32)):
        got = S3.split3_parts(xyz)
        exp = split3_np.parts(xyz)
        assert len(got) == len(exp)
        for a, b in zip(got, exp):
            assert np.array_equal(np.sort(a), np.sort(b))
