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
    CUDA tensors on the same device — nothing travels through the host.  Runs in a process of its own (tests/_chamfer_torch_worker.py):
    torch initialises the device first there, as it does in the reference's training process."""
    import os, subprocess, sys
    from conftest import ROOT, _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_chamfer_torch_worker.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "chamfer device tensors ok" in r.stdout
