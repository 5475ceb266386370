"""Drop-in for the Semantic3D variant's ``chamfer3D.dist_chamfer_3D`` module
(/root/reference/SSRD_AL_semantic3d/chamfer3D/dist_chamfer_3D.py:29-81, imported by fps_gcn_cuda.py:4,33).
``chamfer_3DDist()(xyz1, xyz2) -> (dist1, dist2, idx1, idx2)`` with squared distances, as the reference's forward.
Accepts NumPy arrays or torch tensors (results come back in the same kind); the kernel is libssdr_al.so's
``ssdr_chamfer3d_forward_dev``.  The reference's backward is never reached by SSDR-AL and is not provided."""
import numpy as np

from ssdr_al import _lib
from ssdr_al._lib import DevArray


class chamfer_3DDist:
    def __call__(self, input1, input2):
        return self.forward(input1, input2)

    def forward(self, input1, input2):
        is_torch = hasattr(input1, "detach")
        a = np.ascontiguousarray(input1.detach().cpu().numpy() if is_torch else input1, np.float32)
        b = np.ascontiguousarray(input2.detach().cpu().numpy() if is_torch else input2, np.float32)
        assert a.ndim == 3 and a.shape[2] == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        assert b.ndim == 3 and b.shape[2] == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        d_a, d_b = DevArray.from_host(a), DevArray.from_host(b)
        d1, d2 = DevArray((B, n), np.float32), DevArray((B, m), np.float32)
        i1, i2 = DevArray((B, n), np.int32), DevArray((B, m), np.int32)
        _lib.check(_lib.lib().ssdr_chamfer3d_forward_dev(d_a.ptr, d_b.ptr, B, n, m, d1.ptr, d2.ptr, i1.ptr, i2.ptr, None))
        _lib.sync()
        out = (d1.to_host(), d2.to_host(), i1.to_host(), i2.to_host())
        if is_torch:
            import torch
            out = tuple(torch.from_numpy(o).to(input1.device) for o in out)
        return out
