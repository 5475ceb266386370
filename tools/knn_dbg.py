import sys, numpy as np, ctypes as C, time
sys.path.insert(0,'.'); sys.path.insert(0,'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
hp._front_end(); hp._pyramid(); _lib.sync()
out=(C.c_int32*4)(); _lib.lib().ssdr_knn_status(None, out); print(list(out))
