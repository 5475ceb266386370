"""Worker of tests/test_distributed.py (gpu): the N > 1 code path of bench.py on ONE GPU — RCCL process group of world size 1,
the exchanges running on the library's device buffers and streams — must select exactly what the plain path selects."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from ssdr_al import _lib, pipeline, synthetic
    from ssdr_al.distributed import Comm
    from ssdr_al.helper_tool import ConfigS3DIS
    _lib.check(_lib.lib().ssdr_init(0))

    class Cfg(ConfigS3DIS):
        num_points = 8192
    W = synthetic.init_weights(0)
    rooms = [synthetic.make_room(7100 + i, density=600.0) for i in range(4)]
    mk = lambda: pipeline.HotPath(W, Cfg, select_per_tile=9, labeled_per_tile=4, precision="bf16x3").load_rooms(rooms)
    plain = mk()
    sel0, _ = plain.step()
    comm = Comm(dist, "cuda")
    hp = mk()
    sel1, _ = hp.step(comm)
    pipe = pipeline.Pipelined(mk, 4)
    sel2, _ = pipe.run(3, comm)
    # the k-center selector (BASELINE configuration 4): labelled regions' rows in the all-gather, the global chain seeded with them, the rule on the device
    mkk = lambda: pipeline.HotPath(W, Cfg, select_per_tile=9, labeled_per_tile=4, precision="bf16x3", selector="kcenter").load_rooms(rooms)
    plain_k = mkk(); plain_k.step()
    hk = mkk(); hk.step(comm)
    res = {"plain": [int(x) for x in sel0], "dist": [int(x) for x in sel1], "dist_pipelined": [int(x) for x in sel2],
           "selected_plain": plain.selected, "selected_dist": hp.selected, "lib": _lib.lib_path(),
           "kcenter_plain": plain_k.selected, "kcenter_dist": hk.selected, "kcenter_rule_path": hk.rule_path, "rule_path": hp.rule_path}
    with open(os.environ["SSDR_TEST_OUT"], "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
