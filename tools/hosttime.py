"""Where the host spends a steady-state step: time inside _select_issue (decisions + uploads + launches), inside the other stages' launch
sequences, and blocked in _select_collect.  Development tool."""
import sys, time, collections, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
import os
comm = None
if os.environ.get("SSDR_BENCH_FORCE_DIST"):      # the sharded code path through RCCL at world size 1 (with SSDR_EMULATE_WORLD=N: the FPS load of N ranks)
    import torch, torch.distributed as dist
    for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29563"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")): os.environ.setdefault(k, v)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from ssdr_al.distributed import Comm
    comm = Comm(dist, "cuda")
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(1000 + i, density=5000.0) for i in range(16)]
def mk():
    return pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
pipe = pipeline.Pipelined(mk, 4 if comm is not None else 5)
pipe.run(6, comm, steady=True)
acc = collections.defaultdict(float)
H = pipeline.HotPath
def wrap(name):
    f = getattr(H, name)
    def g(self, *a, **k):
        t = time.perf_counter(); r = f(self, *a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(H, name, g)
for n in ("_front_end", "_pyramid", "_infer", "_score_async", "_select_issue", "_select_collect", "_candidates"):
    wrap(n)
orig_to_host = _lib.DevArray.to_host
def th(self, stream=None):
    t = time.perf_counter(); r = orig_to_host(self, stream); acc["to_host"] += time.perf_counter() - t; return r
_lib.DevArray.to_host = th
N = 60
_lib.sync(); t0 = time.perf_counter(); pipe.run(N, comm, steady=True); _lib.sync(); T = time.perf_counter() - t0
print("step %.3f ms%s" % (T / N * 1e3, "" if comm is None else "  (sharded path, world 1, emulated ranks %s, rule %s)" % (os.environ.get("SSDR_EMULATE_WORLD", "1"), pipe.hp[0].rule_path)))
for k, v in sorted(acc.items(), key=lambda x: -x[1]): print("  %-18s %.3f ms/step" % (k, v / N * 1e3))
