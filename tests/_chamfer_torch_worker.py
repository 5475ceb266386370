"""Worker of tests/test_chamfer3d.py (gpu): chamfer_3DDist on CUDA tensors, produced and consumed on a non-default torch stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))


def _brute(a, b):
    d = b[:, None, :, :] - a[:, :, None, :]
    dist = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return dist.min(2), dist.argmin(2).astype(np.int32)


def main():
    import torch
    assert torch.cuda.is_available(), "torch sees no GPU"
    torch.cuda.set_device(0)
    from chamfer3D.dist_chamfer_3D import chamfer_3DDist
    from ssdr_al import _lib
    rng = np.random.default_rng(6)
    a = rng.random((3, 900, 3), dtype=np.float32); b = rng.random((3, 640, 3), dtype=np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    torch.cuda.synchronize()
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
    # default stream as well
    d1b, _, i1b, _ = chamfer_3DDist()(ta, tb)
    torch.cuda.synchronize()
    assert np.array_equal(d1b.cpu().numpy(), e1) and np.array_equal(i1b.cpu().numpy(), ei1)
    assert _lib.lib_path().endswith("libssdr_al.so")
    print("chamfer device tensors ok")


if __name__ == "__main__":
    main()
