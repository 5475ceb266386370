"""Worker of tests/test_select.py::test_cooperative_fps_reports_a_launch_that_is_not_co_resident: runs in a process of its own because the
grid override is read once per process (SSDR_FPS_COOP_G)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ssdr-al_amd")):
    sys.path.insert(0, p)
from ssdr_al import _lib  # noqa: E402

_lib.use(os.path.join(ROOT, "ssdr-al_amd", "libssdr_al.so"))
L = _lib.lib()
rng = np.random.default_rng(1)
n, D, count = 6000, 16, 40           # D != 32 and n > 4096: the cooperative kernel fps_coop
f = rng.normal(size=(n, D))
d_f = _lib.DevArray.from_host(f)
d_o = _lib.DevArray((count,), np.int32)
_lib.check(L.ssdr_fps_dev(d_f.ptr, n, D, 3, count, d_o.ptr, None))
import ctypes as C
st = C.c_int(0)
rc = L.ssdr_select_status(None, C.byref(st))
out = d_o.to_host()
print("RC", rc, "STATUS", st.value, "MINUS", int((out < 0).sum()), "FIRST", int(out[0]))
if rc == 0:      # a healthy run: the picks must be the reference's sequence
    sys.path.insert(0, ROOT)
    from oracle import select_np
    exp = select_np.farthest_features_sample(f, count, start=3)
    print("MATCH", int(np.array_equal(out, np.asarray(exp))))
