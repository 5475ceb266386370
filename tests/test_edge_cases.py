"""Edge cases of the C ABI: empty / tiny / ragged inputs, unsupported arguments, error reporting."""
import numpy as np
import pytest

from conftest import assert_bits_equal


def test_knn_degenerate_shapes(backend, orc):
    from ssdr_al import _lib, knn
    rng = np.random.default_rng(0)
    p = rng.random((1, 3), dtype=np.float32)
    assert_bits_equal(knn.knn(p, p, 1), orc.knn(p, p, 1))                    # one point
    assert_bits_equal(knn.knn(p, p, 4), orc.knn(p, p, 4))                    # K > npts: trailing zeros
    q = np.zeros((0, 3), np.float32)
    assert knn.knn(p, q, 3).shape == (0, 3)                                  # no queries
    big = rng.random((300, 3), dtype=np.float32)
    assert_bits_equal(knn.knn(big, big[:50], 40), orc.knn(big, big[:50], 40))   # generic-K kernel, K > 16
    assert_bits_equal(knn.knn(big, big[:7], 256), orc.knn(big, big[:7], 256))   # largest supported K
    with pytest.raises(_lib.SsdrError) as e:
        knn.knn(big, big, 257)
    assert e.value.status == 5
    # 11 points: exactly one split above the leaf size of 10
    p11 = rng.random((11, 3), dtype=np.float32)
    assert_bits_equal(knn.knn(p11, p11, 11), orc.knn(p11, p11, 11))
    # collinear / coplanar points (zero spread in two dimensions)
    line = np.zeros((200, 3), np.float32); line[:, 0] = rng.random(200, dtype=np.float32)
    assert_bits_equal(knn.knn(line, line, 16), orc.knn(line, line, 16))


def test_knn_batch_ragged_query_count(backend, orc):
    from ssdr_al import knn
    rng = np.random.default_rng(1)
    P = rng.random((4, 333, 3), dtype=np.float32)
    Q = rng.random((4, 77, 3), dtype=np.float32) * 3 - 1
    assert_bits_equal(knn.knn_batch(P, Q, 16), orc.knn_batch(P, Q, 16))
    assert_bits_equal(knn.knn_batch(P, Q, 1), orc.knn_batch(P, Q, 1))


def test_pyramid_rejects_too_small_tiles(backend):
    from ssdr_al import _lib, randlanet
    from oracle import randla_np as R
    net = randlanet.Network().load(R.init_weights(0))
    x = np.random.default_rng(0).random((1, 256, 3), dtype=np.float32)        # 256/4/4/4/4/2 = 0 points at the last level
    with pytest.raises(_lib.SsdrError) as e:
        net.infer(np.concatenate([x, x], -1), x)
    assert e.value.status == 1


def test_subsample_single_point_and_single_voxel(backend, orc):
    from ssdr_al import subsampling
    one = np.array([[1.5, -2.25, 0.125]], np.float32)
    assert_bits_equal(subsampling.compute(one, sampleDl=0.04), orc.grid_subsampling(one, sampleDl=0.04)[0])
    rng = np.random.default_rng(2)
    pts = (rng.random((5000, 3), dtype=np.float32) * 0.01).astype(np.float32)            # everything in one voxel
    lab = rng.integers(0, 3, 5000).astype(np.int32)
    got = subsampling.compute(pts, classes=lab, sampleDl=1.0)
    exp = orc.grid_subsampling(pts, None, lab, 1.0)
    assert got[0].shape == (1, 3)
    for a, b in zip(got, exp):
        assert_bits_equal(a, b)
    # huge extent with a tiny cell: voxel keys need more than 32 bits (rare high radix passes)
    far = (rng.random((3000, 3), dtype=np.float32) * np.float32(4000.0)).astype(np.float32)
    got = subsampling.compute(far, sampleDl=0.01, order="key")
    exp = orc.grid_subsampling(far, sampleDl=0.01, order="key")[0]
    assert_bits_equal(got, exp)


def test_error_channel_reports_message(backend):
    from ssdr_al import _lib
    L = _lib.lib()
    assert L.ssdr_knn(None, 5, 3, None, 5, 3, None) != 0
    assert b"NULL" in L.ssdr_last_error()
    assert L.ssdr_version().startswith(b"ssdr_al")


def test_first_calls_on_different_streams_from_threads_and_stream_reuse(backend, orc):
    """The reference calls knn_search from several workers: first calls on different streams from different host threads must not race on
    the library's per-stream scratch table, and a destroyed stream's scratch (status tickets, forests) is forgotten."""
    import ctypes as C
    import threading
    from ssdr_al import _lib, knn
    from ssdr_al._lib import DevArray
    L = _lib.lib()
    _lib.check(L.ssdr_init(0))
    rng = np.random.default_rng(5)
    pts = [rng.random((1, 400 + 37 * i, 3), dtype=np.float32) for i in range(4)]
    d_in = [DevArray.from_host(p) for p in pts]
    d_out = [DevArray((1, p.shape[1], 16), np.int32) for p in pts]
    for rounds in range(2):                    # second round: fresh streams (handles may be recycled by the runtime)
        streams = []
        for _ in pts:
            s = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(s))); streams.append(s.value)
        errs = []

        def work(i):
            try:
                _lib.check(L.ssdr_knn_batch_dev(d_in[i].ptr, 1, pts[i].shape[1], 3, d_in[i].ptr, pts[i].shape[1], 16, d_out[i].ptr, streams[i]))
                assert knn.knn_status(streams[i])[2] == 0
            except Exception as e:             # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(pts))]
        [t.start() for t in th]; [t.join() for t in th]
        assert not errs, errs
        for i, p in enumerate(pts):
            assert np.array_equal(d_out[i].to_host(streams[i]), orc.knn_batch(p, p, 16).astype(np.int32))
        for s in streams:
            _lib.check(L.ssdr_stream_destroy(s))
