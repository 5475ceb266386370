"""Worker of tests/test_select.py::test_cooperative_chains_in_flight_on_several_streams: three cooperative FPS chains enqueued on three
streams before anything is waited for (what `bench.py --select-lag 2` does with the sharded run's global chains).  Runs in a process of its own
because the budget override is read once per process (SSDR_FPS_COOP_BUDGET)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ssdr-al_amd")):
    sys.path.insert(0, p)
from oracle import select_np  # noqa: E402
from ssdr_al import _lib  # noqa: E402

_lib.use(os.path.join(ROOT, "ssdr-al_amd", "libssdr_al.so"))
L = _lib.lib()
rng = np.random.default_rng(2)
shapes = [(20000, 32, 300), (9472, 32, 400), (6000, 16, 60)]      # fps_coop_reg (counter form, 40 workgroups), fps_coop_tag (19), fps_coop
streams, feats, outs = [], [], []
for n, D, count in shapes:
    st = C.c_void_p(); _lib.check(L.ssdr_stream_create(C.byref(st))); streams.append(st.value)
    f = rng.normal(size=(n, D)); feats.append(f)
    outs.append((_lib.DevArray.from_host(f), _lib.DevArray((count,), np.int32)))
for rep in range(2):          # twice: the second round meets the first round's events in the account
    for (n, D, count), s, (d_f, d_o) in zip(shapes, streams, outs):
        _lib.check(L.ssdr_fps_dev(d_f.ptr, n, D, 5, count, d_o.ptr, s))
ok = True
for (n, D, count), s, f, (d_f, d_o) in zip(shapes, streams, feats, outs):
    st = C.c_int(0)
    rc = L.ssdr_select_status(s, C.byref(st))
    got = d_o.to_host(s)
    match = bool(np.array_equal(got, np.asarray(select_np.farthest_features_sample(f, count, start=5))))
    print("N", n, "RC", rc, "STATUS", st.value, "MATCH", int(match))
    ok = ok and rc == 0 and st.value == 0 and match
print("ALL", int(ok))
