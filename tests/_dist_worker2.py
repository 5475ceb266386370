"""Worker of tests/test_distributed.py (round 6): one rank of the sharded SELECTION on the CPU logic build + gloo with UNEQUAL shards
(SSDR_TEST_SHARDS = clouds per rank, e.g. "3,2,1") and, optionally, one rank whose clouds hold no labelled region (SSDR_TEST_NOLAB_RANK):
the Bmax / Smax / nl_max paddings of pipeline._dist_setup, a rank that sends no labelled row, both selectors — against ONE process over the
union of the clouds.  The clouds come with fabricated network outputs (HotPath.from_clouds): the stages in front of the scoring shard with
no collective and are covered by the two-rank test."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from _fabricate import make_clouds
    from ssdr_al import _lib, pipeline
    from ssdr_al.distributed import Comm
    from ssdr_al.helper_tool import ConfigS3DIS
    _lib.use(os.path.join(ROOT, "tests", "hipemu", "libssdr_al_emu.so"))
    shards = [int(x) for x in os.environ["SSDR_TEST_SHARDS"].split(",")]
    assert len(shards) == world
    nolab = int(os.environ.get("SSDR_TEST_NOLAB_RANK", "-1"))
    first = [sum(shards[:r]) for r in range(world + 1)]
    clouds, labelled, sel_list = make_clouds(77, first[-1], (40, 70), 6, 30, labelled_per_cloud=6)
    if nolab >= 0:
        for b in range(first[nolab], first[nolab + 1]):
            labelled[b] = set()
    mine = list(range(first[rank], first[rank + 1]))
    comm = Comm(dist, "cpu")
    res = {"rank": rank}
    kw = dict(sampler_args=("sb", "WetSU", "clsbal", "gcn_fps"), gcn_number=1, gcn_top=0, min_size=8, round_num=3, label_seed=31, select_per_tile=7)
    for selector in ("fps", "kcenter"):
        hp = pipeline.HotPath.from_clouds([clouds[i] for i in mine], [labelled[i] for i in mine], sel_list, ConfigS3DIS, room_ids=mine, selector=selector, **kw)
        hp.step_selection(comm)
        res[selector] = hp.selected; res[selector + "_path"] = hp.rule_path
        hp.set_labeled(hp.labeled)         # relabelling (here: the same sets) drops the sharded tables; the next step rebuilds them
        hp.step_selection(comm)
        res[selector + "_again"] = hp.selected
        if rank == 0:
            one = pipeline.HotPath.from_clouds(clouds, labelled, sel_list, ConfigS3DIS, selector=selector, **kw)
            one.step_selection()
            res[selector + "_single"] = one.selected
    res["n_lab_mine"] = int(sum(len(v) for v in hp.lab_rows.values()))
    with open(os.path.join(os.environ["SSDR_TEST_OUT"], "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
