"""Exchange steps of the sharded path.  Rooms/tiles are sharded across ranks (one process per
GPU); the data path has no collective until selection, where three small exchanges make the result identical to a
single-process run over the union of the rooms:
  1. all-reduce of the 64-bin dominant-class histogram (class balance, sampler2.py:262-266),
  2. all-gather of the per-superpoint region uncertainties (global top-`batch` cut, sampler2.py:640, :533-552),
  3. all-gather of the candidates' propagated features before the replicated global FPS (fps_gcn_cpu.py:169-170).
torch.distributed backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np


class Comm:
    def __init__(self, dist, device):
        import torch
        self.dist, self.device, self.torch = dist, device, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def allgather_var(self, a):
        """a [n, ...] -> (concatenation over ranks in rank order, per-rank counts)."""
        torch = self.torch
        a = np.ascontiguousarray(a)
        n = torch.tensor([a.shape[0]], device=self.device, dtype=torch.int64)
        ns = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(ns, n)
        counts = [int(x[0]) for x in ns]
        pad = torch.zeros((max(max(counts), 1),) + a.shape[1:], device=self.device, dtype=torch.from_numpy(a).dtype)
        if a.shape[0]:
            pad[: a.shape[0]] = torch.from_numpy(a).to(self.device)
        bufs = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)                      # padded all-gather (variable counts)
        out = torch.cat([b[:c] for b, c in zip(bufs, counts)]).cpu().numpy()
        return np.ascontiguousarray(out), counts
