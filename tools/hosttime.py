"""Where the host spends a steady-state step: time inside _select_issue (decisions + uploads + launches), inside the other stages' launch
sequences, and blocked in _select_collect.  Development tool."""
import sys, time, collections, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(1000 + i, density=5000.0) for i in range(16)]
def mk():
    return pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
pipe = pipeline.Pipelined(mk, 5)
pipe.run(6, None, steady=True)
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
_lib.sync(); t0 = time.perf_counter(); pipe.run(N, None, steady=True); _lib.sync(); T = time.perf_counter() - t0
print("step %.3f ms" % (T / N * 1e3))
for k, v in sorted(acc.items(), key=lambda x: -x[1]): print("  %-18s %.3f ms/step" % (k, v / N * 1e3))
