"""N > 1 path on the CPU: two gloo ranks, each running the sharded hot path (CPU logic build of the kernels) on its
own rooms.  The three selection exchanges (class histogram all-reduce, region-uncertainty all-gather, candidate
feature all-gather + replicated FPS) must make the sharded result identical to ONE process over all rooms."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_two_rank_sharded_selection_equals_single_process(tmp_path, emu_lib):
    env = dict(os.environ, SSDR_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_worker.py")]
    subprocess.run(cmd, check=True, env=env, timeout=1200, cwd=ROOT)
    r = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    assert r[0]["n_all"] == r[1]["n_all"] and r[0]["sel"] == r[1]["sel"] == r[0]["expect"] == r[1]["expect"]
    assert r[0]["selected"] == r[1]["selected"]
    assert len(r[0]["selected"]) == 20 and len(set(map(tuple, r[0]["selected"]))) == 20
    assert {c for c, _ in r[0]["selected"]} <= {0, 1, 2, 3}
    assert r[0]["selected"] == r[0]["single"]          # sharded == single process, index for index
    # the sharded FPS selection took the device-side rule (no ranking read-back, no NumPy between the exchanges), and the host rule agrees with it
    assert all(x["rule_path"] == ["sharded-device", "host"] and x["host_rule_equal"] for x in r)
    assert r[0]["kcenter"] == r[1]["kcenter"] == r[0]["single_kcenter"] and len(r[0]["kcenter"]) == 20      # global k-center (configuration 4)
    # ... through the device-side rule as well (no read-back, no NumPy between the exchanges), and the host-side rule picks the same regions
    assert all(x["kcenter_rule_path"] == ["sharded-device", "host"] and x["kcenter_host_equal"] for x in r)
    assert all(x["pipelined_equal"] and x["pipelined_selected"] == x["selected"] for x in r)    # batches in flight: same result


@pytest.mark.parametrize("shards,nolab,port", [("3,2,1", 2, 29551), ("1,1,1,1,1,1,1,1", -1, 29553)])
def test_unequal_shards_and_eight_ranks_equal_single_process(tmp_path, emu_lib, shards, nolab, port):
    """world 3 with 3 + 2 + 1 clouds (the Bmax / Smax / nl_max paddings, one rank without a labelled region) and world 8 with one cloud each:
    the sharded selection (device-side rule, both selectors, labels re-set between two steps) == ONE process over the union of the clouds,
    index for index.  Clouds with fabricated network outputs: the stages in front of the scoring have no collective."""
    world = len(shards.split(","))
    env = dict(os.environ, SSDR_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="2" if world <= 3 else "1", SSDR_TEST_SHARDS=shards, SSDR_TEST_NOLAB_RANK=str(nolab))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker2.py")]
    subprocess.run(cmd, check=True, env=env, timeout=2400, cwd=ROOT)
    r = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(world)]
    rooms = sum(int(x) for x in shards.split(","))
    for sel in ("fps", "kcenter"):
        assert all(x[sel] == r[0][sel] == x[sel + "_again"] for x in r)
        assert r[0][sel] == r[0][sel + "_single"] and len(r[0][sel]) == 7 * rooms
        assert all(x[sel + "_path"] == "sharded-device" for x in r)
    if nolab >= 0:
        assert r[nolab]["n_lab_mine"] == 0 and all(x["n_lab_mine"] > 0 for i, x in enumerate(r) if i != nolab)


@pytest.mark.gpu
def test_rccl_exchange_path_on_one_gpu_equals_plain_path(tmp_path):
    """bench.py's N > 1 code path (device-resident exchanges through RCCL, ordered on the library's streams) on one GPU."""
    from conftest import _have_gpu
    if not _have_gpu():
        pytest.skip("no GPU")
    out = tmp_path / "res.json"
    env = dict(os.environ, SSDR_TEST_OUT=str(out), MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py")], check=True, env=env, timeout=600, cwd=ROOT)
    r = json.load(open(out))
    assert len(r["plain"]) == 36 and r["plain"] == r["dist"] == r["dist_pipelined"]
    assert r["selected_plain"] == r["selected_dist"]
    assert r["rule_path"] == "sharded-device" and r["kcenter_rule_path"] == "sharded-device"
    assert len(r["kcenter_plain"]) == 36 and r["kcenter_plain"] == r["kcenter_dist"]      # the k-center selector through the exchange path
    assert r["lib"].endswith("libssdr_al.so")
