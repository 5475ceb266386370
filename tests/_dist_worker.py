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
    from ssdr_al.distributed import Comm
    from ssdr_al.helper_tool import ConfigS3DIS
    _lib.use(os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu.so"))

    class Cfg(ConfigS3DIS):
        num_points = 1024
    per = 2
    W = R.init_weights(0)
    all_rooms = [synthetic.make_room(7000 + i, density=80.0) for i in range(per * world)]
    mine = list(range(rank * per, (rank + 1) * per))
    hp = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2).load_rooms([all_rooms[i] for i in mine], mine)
    comm = Comm(dist, "cpu")
    sel, unl = hp.step(comm)
    path = hp.rule_path
    os.environ["SSDR_SELECT_HOST_RULE"] = "1"      # the round-2 formulation (ranking read back, rule in NumPy): must pick the same regions
    hsel, _ = hp.step(comm)
    host_path, host_selected = hp.rule_path, hp.selected
    del os.environ["SSDR_SELECT_HOST_RULE"]
    sel, unl = hp.step(comm)
    # the same batches kept in flight on separate streams, exchanges included (what bench.py runs for N > 1)
    pipe = pipeline.Pipelined(lambda: pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2).load_rooms([all_rooms[i] for i in mine], mine), 2)
    psel, _ = pipe.run(2, comm)
    # BASELINE configuration 4: all-gather of the candidates' AND the labelled regions' features, then the global k-center (replicated)
    hk = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2, selector="kcenter").load_rooms([all_rooms[i] for i in mine], mine)
    hk.step(comm)
    kc_sharded, kc_path = hk.selected, hk.rule_path
    os.environ["SSDR_SELECT_HOST_RULE"] = "1"      # ... and the k-center selector through the host-side rule: the same regions
    hk.step(comm)
    kc_host, kc_host_path = hk.selected, hk.rule_path
    del os.environ["SSDR_SELECT_HOST_RULE"]
    res = {"rule_path": [path, host_path], "kcenter_rule_path": [kc_path, kc_host_path], "kcenter_host_equal": kc_host == kc_sharded, "host_rule_equal": bool(np.array_equal(hsel, sel) and host_selected == hp.selected), "kcenter": kc_sharded, "pipelined_equal": bool(np.array_equal(psel, sel)), "pipelined_selected": pipe.hp[1].selected,"rank": rank, "sel": [int(x) for x in sel], "selected": hp.selected, "n_all": int(len(hp.comb_all)),
           "expect": [int(x) for x in S.farthest_features_sample(hp.comb_all, len(sel), 0)]}
    if rank == 0:      # the same job in ONE process over the union of the rooms
        one = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2).load_rooms(all_rooms, list(range(per * world)))
        one.step()
        res["single"] = one.selected
        onek = pipeline.HotPath(W, Cfg, select_per_tile=5, labeled_per_tile=2, selector="kcenter").load_rooms(all_rooms, list(range(per * world)))
        onek.step()
        res["single_kcenter"] = onek.selected
    with open(os.path.join(os.environ["SSDR_TEST_OUT"], "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
