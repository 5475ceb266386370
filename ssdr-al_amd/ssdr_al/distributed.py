"""The one exchange step of the sharded path: tiles/rooms are sharded across ranks with no
data-path collective; before the global FPS each rank's propagated candidate features are all-gathered
(RCCL over xGMI on GPUs = torch.distributed backend "nccl"; "gloo" in the CPU tests) and FPS runs replicated."""
import numpy as np


def make_gather(dist, device):
    """Returns gather(comb_local f64[n,32], batch_local) -> (comb_all f64[sum n,32] in rank order, sum batch)."""
    import torch
    world = dist.get_world_size()

    def gather(comb_local, batch_local):
        n = torch.tensor([len(comb_local), batch_local], device=device, dtype=torch.int64)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        nmax = int(max(int(x[0]) for x in ns))
        pad = torch.zeros((nmax, comb_local.shape[1]), device=device, dtype=torch.float64)
        pad[: len(comb_local)] = torch.from_numpy(np.ascontiguousarray(comb_local)).to(device)
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad)                      # padded all-gather (variable counts)
        comb = torch.cat([b[: int(x[0])] for b, x in zip(bufs, ns)]).cpu().numpy()
        return np.ascontiguousarray(comb), int(sum(int(x[1]) for x in ns))
    return gather
