"""Worker of tests/test_distributed.py: one rank of the sharded hot path on the CPU logic build + gloo."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from oracle import randla_np as R
    from oracle import select_np as S
    from ssdr_al import _lib, pipeline, synthetic
    from ssdr_al.distributed import make_gather
    from ssdr_al.helper_tool import ConfigS3DIS
    _lib.use(os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu.so"))

    class Cfg(ConfigS3DIS):
        num_points = 2048
    rooms = [synthetic.make_room(7000 + rank * 2 + i, density=150.0) for i in range(2)]
    hp = pipeline.HotPath(R.init_weights(0), Cfg, select_per_tile=5, labeled_per_tile=2).load_rooms(rooms)
    seen = {}
    inner = make_gather(dist, "cpu")

    def gather(comb, batch):
        out = inner(comb, batch)
        seen["local"], seen["all"], seen["batch"] = comb.copy(), out[0].copy(), out[1]
        return out
    sel, unl = hp.step(gather)
    expect = S.farthest_features_sample(seen["all"], seen["batch"], 0)
    res = {"rank": rank, "sel": [int(x) for x in sel], "expect": [int(x) for x in expect], "n_local": len(seen["local"]),
           "n_all": len(seen["all"]), "batch": seen["batch"], "local_sum": float(seen["local"].sum()), "all_sum": float(seen["all"].sum())}
    with open(os.path.join(os.environ["SSDR_TEST_OUT"], "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
