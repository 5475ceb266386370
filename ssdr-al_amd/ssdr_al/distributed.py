"""Exchange steps of the sharded path.  Rooms/tiles are sharded across ranks (one process per
GPU); the data path has no collective until selection, where three small exchanges make the result identical to a
single-process run over the union of the rooms:
  1. all-reduce of the 64-bin dominant-class histogram (class balance, sampler2.py:262-266),
  2. all-gather of the per-superpoint region uncertainties (global top-`batch` cut, sampler2.py:640, :533-552),
  3. all-gather of the candidates' propagated features before the replicated global FPS (fps_gcn_cpu.py:169-170).
torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

The collectives run directly on the library's device buffers (``DevArray.__cuda_array_interface__`` -> zero-copy torch
views) and are ordered on the caller's HIP stream (``torch.cuda.ExternalStream``): RCCL's stream waits for what the stream
holds so far, and the stream continues after the collective.  Nothing is staged through the host and the host never waits."""
import contextlib

import numpy as np

from . import _lib


class Comm:
    def __init__(self, dist, device):
        import torch
        self.dist, self.device, self.torch = dist, device, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.cuda = str(device).startswith("cuda")
        self._ext = {}

    def view(self, arr):
        """torch tensor sharing the memory of a DevArray (no copy)."""
        if self.cuda:
            return self.torch.as_tensor(arr, device=self.device)
        return self.torch.from_numpy(arr.host_view())              # CPU logic build: device memory is host memory

    def on_stream(self, stream):
        """Context in which collectives are ordered on the library stream `stream` (None = the library's main stream)."""
        if not self.cuda:
            return contextlib.nullcontext()                         # CPU logic build: every launch has finished when it returns
        import ctypes as C
        if stream is None:
            p = C.c_void_p()
            _lib.check(_lib.lib().ssdr_main_stream(C.byref(p)))
            stream = p.value
        key = int(stream)
        if key not in self._ext:
            self._ext[key] = self.torch.cuda.ExternalStream(key)
        return self.torch.cuda.stream(self._ext[key])

    def allreduce_sum_(self, arr, stream=None):
        """In-place sum over ranks of a DevArray."""
        with self.on_stream(stream):
            self.dist.all_reduce(self.view(arr))

    def allgather_(self, arr_in, arr_out, stream=None):
        """arr_out [world, *arr_in.shape] <- arr_in of every rank, in rank order (equal sizes: callers pad)."""
        with self.on_stream(stream):
            self.dist.all_gather_into_tensor(self.view(arr_out).reshape(-1), self.view(arr_in).reshape(-1)) if self.cuda else \
                self.dist.all_gather(list(self.view(arr_out).reshape(self.world, -1).unbind(0)), self.view(arr_in).reshape(-1))

    def allgather_host(self, a):
        """Setup-time helper (untimed): equal-shaped NumPy arrays of every rank, stacked in rank order."""
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return np.stack([o.cpu().numpy() for o in out])
