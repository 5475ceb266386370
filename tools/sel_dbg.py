"""sizes of the chamfer problem of one bench step (candidates per cloud, points per candidate superpoint)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'ssdr-al_amd')
from ssdr_al import _lib, pipeline, synthetic
from ssdr_al.helper_tool import ConfigS3DIS
_lib.check(_lib.lib().ssdr_init(0))
W = synthetic.init_weights(0)
rooms = [synthetic.make_room(5000 + i, density=5000.0) for i in range(16)]
hp = pipeline.HotPath(W, ConfigS3DIS, precision="bf16x3").load_rooms(rooms)
orig = _lib.lib().ssdr_cloud_graph_batch_dev
hp.step()
off = hp.sp_off.to_host()
sizes_all = np.diff(off)
print("superpoints", len(sizes_all), "size mean %.1f median %d max %d" % (sizes_all.mean(), np.median(sizes_all), sizes_all.max()))
sp = np.asarray(hp.unl_sp); cl = np.asarray(hp.unl_cloud_ids)
print("candidates", len(sp), "clouds", len(set(cl.tolist())))
base = np.asarray(hp.sp_base, np.int64)
g = sp + base[cl] if hasattr(hp, "sp_base") else sp
cs = sizes_all[g]
print("candidate size mean %.1f median %d max %d  <=64: %d  <=128: %d  >128: %d" % (cs.mean(), np.median(cs), cs.max(), (cs <= 64).sum(), (cs <= 128).sum(), (cs > 128).sum()))
ev = 0
for c in sorted(set(cl.tolist())):
    s = cs[cl == c]; ev += s.sum() ** 2
    print(" cloud", c, "n", len(s), "points", int(s.sum()))
print("point pairs %.3e" % ev)
