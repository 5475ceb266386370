"""Do two FPS chains (the cooperative kernel of the sharded run's sizes) overlap when enqueued on the pipeline's two selection streams — the library
stream and the stream created last, after the spare and the four stage streams (pipeline.Pipelined)?  And on two streams created back to back?"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib
from ssdr_al._lib import DevArray
L = _lib.lib(); _lib.check(L.ssdr_init(0))
n, count = int(sys.argv[1]) if len(sys.argv) > 1 else 4736, int(sys.argv[2]) if len(sys.argv) > 2 else 2368
def mk():
    st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); return st.value
streams = [mk() for _ in range(8)]          # spare, front, knn, infer, score, sel2, + two more
f = DevArray.from_host(np.random.default_rng(1).normal(size=(n, 32)))
outs = [DevArray((count,), np.int32) for _ in range(2)]
def run(pair, label):
    for rep in range(2):
        _lib.sync(); [ _lib.sync(s) for s in streams ]
        t0 = time.perf_counter()
        for o, st in zip(outs, pair):
            _lib.check(L.ssdr_fps_dev(f.ptr, n, 32, 0, count, o.ptr, st))
        _lib.sync(); [ _lib.sync(s) for s in streams ]
        dt = (time.perf_counter() - t0) * 1e3
    print("%-44s %.2f ms for two chains" % (label, dt))
_lib.check(L.ssdr_fps_dev(f.ptr, n, 32, 0, count, outs[0].ptr, None)); _lib.sync()
t0 = time.perf_counter(); _lib.check(L.ssdr_fps_dev(f.ptr, n, 32, 0, count, outs[0].ptr, None)); _lib.sync()
print("one chain: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
run((None, None), "both on the library stream")
run((None, streams[5]), "library stream + the pipeline's own (6th)")
for k in range(8):
    run((None, streams[k]), "library stream + created stream #%d" % k)
