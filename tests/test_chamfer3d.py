"""chamfer_3DDist (Semantic3D variant's CUDA op): PARITY UNPINNED by the reference (CUDA cannot run here); checked
against a float32 NumPy brute force with the same expression order and the same lowest-index tie rule."""
import numpy as np


def _brute(a, b):
    d = b[:, None, :, :] - a[:, :, None, :]                       # [B,n,m,3], dx = b - a as in chamfer3D.cu:34-36
    dist = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return dist.min(2), dist.argmin(2).astype(np.int32)


def test_chamfer3d_forward(backend):
    from chamfer3D.dist_chamfer_3D import chamfer_3DDist          # fps_gcn_cuda.py:4
    rng = np.random.default_rng(4)
    n, m = (700, 530) if backend == "emu" else (1000, 777)
    a = rng.random((2, n, 3), dtype=np.float32); b = rng.random((2, m, 3), dtype=np.float32)
    b[0, 5] = b[0, 300]                                           # exact tie: lowest index must win
    a[1, :10] = b[1, :10]
    d1, d2, i1, i2 = chamfer_3DDist()(a, b)
    e1, ei1 = _brute(a, b); e2, ei2 = _brute(b, a)
    assert np.array_equal(d1, e1) and np.array_equal(i1, ei1)
    assert np.array_equal(d2, e2) and np.array_equal(i2, ei2)
    assert (d1[1, :10] == 0).all()
    # create_cd_cuda's use of it (fps_gcn_cuda.py:26-27): mean(sqrt(d1)) + mean(sqrt(d2))
    cd = np.sqrt(d1[0]).mean() + np.sqrt(d2[0]).mean()
    assert np.isfinite(cd) and cd > 0


import pytest


@pytest.mark.gpu
def test_chamfer3d_forward_on_device_tensors():
    """The reference's contract is GPU tensors only (dist_chamfer_3D.py:30): CUDA tensors go in by their device pointers, the results are
    CUDA tensors on the same device — nothing travels through the host."""
    from conftest import GPU_LIB, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    from ssdr_al import _lib
    _lib.use(GPU_LIB)
    try:
        from chamfer3D.dist_chamfer_3D import chamfer_3DDist
        rng = np.random.default_rng(6)
        a = rng.random((3, 900, 3), dtype=np.float32); b = rng.random((3, 640, 3), dtype=np.float32)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            d1, d2, i1, i2 = chamfer_3DDist()(ta * 1.0, tb * 1.0)         # inputs produced on the same (non-default) stream
            cd = torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1)           # create_cd_cuda's use (fps_gcn_cuda.py:26-27), consumed on that stream
        s.synchronize()
        assert d1.is_cuda and i1.is_cuda and i1.dtype == torch.int32
        e1, ei1 = _brute(a, b); e2, ei2 = _brute(b, a)
        assert np.array_equal(d1.cpu().numpy(), e1) and np.array_equal(i1.cpu().numpy(), ei1)
        assert np.array_equal(d2.cpu().numpy(), e2) and np.array_equal(i2.cpu().numpy(), ei2)
        assert np.allclose(cd.cpu().numpy(), np.sqrt(e1).mean(1) + np.sqrt(e2).mean(1), rtol=1e-6)
    finally:
        _lib.use(None)
