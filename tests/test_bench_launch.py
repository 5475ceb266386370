"""bench.py --gpus N without a launcher starts N ranks itself (one process per GPU, torch.distributed.run as a child process)
and rank 0 prints ONE JSON line with n_gpus == N.  Here: the CPU logic build of the kernels + gloo (bench.py --emu, a test-only
mode that measures nothing) — the launcher, the rank / world bookkeeping and the N > 1 control flow of the bench are what is tested."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_gpus2_launches_two_ranks(emu_lib):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--emu", "--pipeline-depth", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    j = _line(r.stdout)
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]      # ONE line on stdout: whatever else the ranks (or a library in them) print goes to stderr
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["ms_per_step"] > 0
    assert j["config"]["selected_per_step"] == 2 * 2 * 5          # both ranks' tiles take part in the global selection
    # the second reading of an N-rank job: the reference's ONE batch_size per round (what one rank's tiles select), named beside the default
    assert j["fixed_batch"]["selected_per_step"] == 2 * 5 and j["fixed_batch"]["ms_per_step"] > 0 and "N^2" in j["config"]["selection_batch"]


def test_bench_refuses_more_ranks_than_gpus():
    """`bench.py --gpus N` on a node with fewer than N GPUs exits non-zero with one line, before any rank is started or any GPU touched
    (the KFD topology is read, not HIP).  This box lists fewer than 64 GPUs whatever it is."""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("SSDR_BENCH_SKIP_DEVICE_COUNT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 64 but this node lists" in (r.stderr + r.stdout)
    assert '{"metric"' not in r.stdout


def test_bench_rejects_world_size_mismatch(emu_lib):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emu"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_failing_rank_fails_the_launch(emu_lib):
    env = dict(os.environ, SSDR_AL_BENCH_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--emu", "--pipeline-depth", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
