#!/usr/bin/env python3
"""the selection half of an AL round at the reference's scale, for a kernel trace: ALRound.run() twice (the second one is the one to read)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ssdr-al_amd"))
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
ar = pipeline.ALRound(W, rooms, 17, ConfigS3DIS, batch_size=10000, precision="bf16x3", selector=sys.argv[1] if len(sys.argv) > 1 else "fps")
for _ in range(2):
    ar.run()
    _lib.sync()
