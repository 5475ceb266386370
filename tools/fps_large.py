#!/usr/bin/env python3
"""FPS at the reference's real scale (ssdr_main_S3DIS2.py:134: 10 000 picks over ~2 x 10^4 candidate regions, 32-d features):
   python tools/fps_large.py [n] [count] [--oracle] [--save f.npy] [--cmp f.npy]
   -> time of ssdr_fps_dev (the form is the library's choice, or SSDR_FPS_COOP_SWEEP / SSDR_FPS_COOP_COUNTER), and the index sequence against the
   NumPy oracle (--oracle) or against a sequence an earlier run saved (--cmp)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
import numpy as np
from ssdr_al import _lib
from ssdr_al._lib import DevArray
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = sys.argv[1:]
n = int(args[0]) if len(args) > 0 else 20000
count = int(args[1]) if len(args) > 1 else 10000
def opt(name):
    return flags[flags.index(name) + 1] if name in flags else None
_lib.check(_lib.lib().ssdr_init(0))
f = np.random.default_rng(1).normal(size=(n, 32))
d_f = DevArray.from_host(f); d_o = DevArray((count,), np.int32)
best = 1e9
for rep in range(3):
    _lib.sync(); t0 = time.perf_counter()
    _lib.check(_lib.lib().ssdr_fps_dev(d_f.ptr, n, 32, 0, count, d_o.ptr, None))
    _lib.sync(); dt = time.perf_counter() - t0
    best = min(best, dt)
st = _lib.lib().ssdr_select_status(None, None)
print("ssdr_fps_dev n=%d count=%d form=%s/%s: %.2f ms (%.3f us per pick) status=%d" % (n, count, os.environ.get("SSDR_FPS_COOP_SWEEP", "-"), os.environ.get("SSDR_FPS_COOP_COUNTER", "-"),
                                                                                best * 1e3, best * 1e6 / count, st), flush=True)
got = d_o.to_host()
if opt("--save"):
    np.save(opt("--save"), got)
if opt("--cmp"):
    print("   sequence identical to %s: %s" % (opt("--cmp"), bool(np.array_equal(got, np.load(opt("--cmp"))))), flush=True)
if "--oracle" in flags:
    from oracle import select_np as S
    t0 = time.perf_counter(); exp = np.asarray(S.farthest_features_sample(f, count, 0)); t1 = time.perf_counter() - t0
    print("   oracle (NumPy): %.1f s; sequences identical: %s" % (t1, bool(np.array_equal(got, exp))), flush=True)
