"""Drop-in for the Semantic3D variant's ``chamfer3D.dist_chamfer_3D`` module
(/root/reference/SSRD_AL_semantic3d/chamfer3D/dist_chamfer_3D.py:29-81, imported by fps_gcn_cuda.py:4,33).
``chamfer_3DDist()(xyz1, xyz2) -> (dist1, dist2, idx1, idx2)`` with squared distances, as the reference's forward.

The reference's contract is "GPU tensors only" (dist_chamfer_3D.py:30): a torch tensor that lives on the GPU is handed to
``ssdr_chamfer3d_forward_dev`` by its device pointer (``data_ptr()``), the four results are allocated by torch on the same device and the
kernel runs on torch's current stream — no host round trip.  NumPy arrays and CPU tensors (tests, callers without a framework) are copied
to the device and back.  The reference's backward is never reached by SSDR-AL and is not provided."""
import numpy as np

from ssdr_al import _lib
from ssdr_al._lib import DevArray


class chamfer_3DDist:
    def __call__(self, input1, input2):
        return self.forward(input1, input2)

    def forward(self, input1, input2):
        is_torch = hasattr(input1, "detach")
        for x in (input1, input2):
            assert len(x.shape) == 3 and x.shape[2] == 3, "Wrong last dimension for the chamfer distance 's input! Check with .size()"
        if is_torch and input1.is_cuda:
            import torch
            a = input1.detach().contiguous().float(); b = input2.detach().to(a.device).contiguous().float()
            B, n, m = a.shape[0], a.shape[1], b.shape[1]
            d1 = torch.zeros(B, n, device=a.device); d2 = torch.zeros(B, m, device=a.device)
            i1 = torch.zeros(B, n, dtype=torch.int32, device=a.device); i2 = torch.zeros(B, m, dtype=torch.int32, device=a.device)
            with torch.cuda.device(a.device):
                stream = torch.cuda.current_stream().cuda_stream or None      # 0 = the legacy default stream: the library's own stream then, joined below
                if stream is None:
                    torch.cuda.current_stream().synchronize()                 # inputs produced on the default stream are complete
                _lib.check(_lib.lib().ssdr_chamfer3d_forward_dev(a.data_ptr(), b.data_ptr(), B, n, m, d1.data_ptr(), d2.data_ptr(), i1.data_ptr(), i2.data_ptr(), stream))
                if stream is None:
                    _lib.sync()
            return d1, d2, i1, i2
        a = np.ascontiguousarray(input1.detach().cpu().numpy() if is_torch else input1, np.float32)
        b = np.ascontiguousarray(input2.detach().cpu().numpy() if is_torch else input2, np.float32)
        B, n, m = a.shape[0], a.shape[1], b.shape[1]
        d_a, d_b = DevArray.from_host(a), DevArray.from_host(b)
        d1, d2 = DevArray((B, n), np.float32), DevArray((B, m), np.float32)
        i1, i2 = DevArray((B, n), np.int32), DevArray((B, m), np.int32)
        _lib.check(_lib.lib().ssdr_chamfer3d_forward_dev(d_a.ptr, d_b.ptr, B, n, m, d1.ptr, d2.ptr, i1.ptr, i2.ptr, None))
        _lib.sync()
        out = (d1.to_host(), d2.to_host(), i1.to_host(), i2.to_host())
        if is_torch:
            import torch
            out = tuple(torch.from_numpy(o) for o in out)
        return out
