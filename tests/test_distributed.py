"""N > 1 path on the CPU: two gloo ranks, each running the sharded hot path (CPU logic build of the kernels) on its
own rooms, all-gathering the candidates' propagated features and running the global FPS replicated."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_two_rank_sharded_selection(tmp_path, emu_lib):
    env = dict(os.environ, SSDR_TEST_OUT=str(tmp_path), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_worker.py")]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=ROOT)
    r = [json.load(open(tmp_path / ("rank%d.json" % i))) for i in range(2)]
    # every rank sees the same gathered set (rank order) and makes the same global selection
    assert r[0]["n_all"] == r[0]["n_local"] + r[1]["n_local"] == r[1]["n_all"]
    assert abs(r[0]["all_sum"] - (r[0]["local_sum"] + r[1]["local_sum"])) < 1e-6 * max(1.0, abs(r[0]["all_sum"]))
    assert r[0]["batch"] == r[1]["batch"] == 20
    assert r[0]["sel"] == r[1]["sel"] == r[0]["expect"] == r[1]["expect"]
    assert len(set(r[0]["sel"])) == len(r[0]["sel"]) and max(r[0]["sel"]) < r[0]["n_all"]
