"""Do the selection chains of two batches overlap when they are enqueued on two streams?  (development tool)"""
import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
L = _lib.lib()
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
hps = [pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms) for _ in range(2)]
for h in hps:
    h.step()
def mk():
    st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); return st.value
for nstreams in (1, 2, 3):
    streams = [None] + [mk() for _ in range(nstreams - 1)] if nstreams > 1 else [None]
    hs = [pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms) for _ in range(nstreams)] if nstreams > 2 else hps[:nstreams]
    for h in hs: h.step()
    for i, h in enumerate(hs): h.sel_stream = streams[i]; h.pipelined = True
    _lib.sync()
    n = 12
    t0 = time.perf_counter()
    for k in range(n):
        h = hs[k % nstreams]
        if k >= nstreams: h._select_collect()
        h._select_issue(None)
    for h in hs: h._select_collect()
    _lib.sync()
    print("%d selection stream(s): %.3f ms per selection" % (nstreams, (time.perf_counter() - t0) / n * 1e3))
