import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn
from cilantro_amd.icp import Context
n = 10_000_000
d = syn.make_pair(n, n, with_normals=True)
rng = np.random.default_rng(3)
Ti = np.linalg.inv(d["T_true"].astype(np.float64))
src = (rng.random((n, 3), dtype=np.float32).astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
ctx = Context(); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(src)
ctx.find_correspondences(d["T_true"].astype(np.float32), float(d["max_sq_dist"]), count=False)
print("deferred queries, whole tiles:", ctx.debug_counters())
idx, d2 = ctx.get_nn()
h = d["h"]
m = idx != capi.NONE_IDX
print("matched", m.sum(), "mean d/h", np.sqrt(d2[m]).mean() / h, "frac d > h", (np.sqrt(d2[m]) > h).mean(), "frac d>0.5h", (np.sqrt(d2[m]) > 0.5 * h).mean(), "grid cell/h", ctx.grid_info().cell / h)
