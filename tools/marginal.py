"""What each stage costs the pipelined step: steady-state ms/step with one stage's launches skipped after warm-up (its buffers keep
the last results, so the later stages still have valid inputs).  Development tool: python tools/marginal.py [skip ...]"""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
def mk():
    return pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
pipe = pipeline.Pipelined(mk, 5)
pipe.run(6, None, steady=True)
def timed(n=40):
    _lib.sync(); t0 = time.perf_counter(); pipe.run(n, None, steady=True); _lib.sync()
    for st in pipe.streams: _lib.sync(st)
    return (time.perf_counter() - t0) / n * 1e3
base = timed()
print("full step: %.3f ms" % base)
H = pipeline.HotPath
orig = {n: getattr(H, n) for n in ("_front_end", "_pyramid", "_infer", "_score_async", "_select_issue", "_select_collect")}
last = {}
def skip(names):
    for n in orig: setattr(H, n, orig[n])
    for n in names:
        if n == "_select":
            def si(self, comm=None): pass
            def sc(self): return last.get("sel")
            H._select_issue, H._select_collect = si, sc
        elif n == "_score_async":
            setattr(H, n, lambda self, comm=None: None)
        else:
            setattr(H, n, lambda self: None)
for names in (["_front_end"], ["_pyramid"], ["_infer"], ["_score_async"], ["_select"], ["_front_end", "_pyramid"], ["_infer", "_score_async"],
              ["_front_end", "_pyramid", "_infer", "_score_async"], ["_front_end", "_pyramid", "_infer", "_score_async", "_select"]):
    skip(names)
    t = timed()
    print("without %-55s %.3f ms   (saves %.3f)" % (" ".join(names), t, base - t))
skip([])
print("full again: %.3f ms" % timed())
