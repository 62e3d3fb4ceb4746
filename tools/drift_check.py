"""Dev tool: cost of the tiled search when the transform has drifted from the sort-time one by a fraction of a cell
in every axis (tiles whose region outgrows the LDS budget go to the clean-up pass)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import Context

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
d = syn.make_pair(n, n, with_normals=False, perturb=0.0)
ctx = Context(0)
ctx.set_target(d["dst"], None); ctx.set_source(d["src"])
ctx.set_option("tiled", 2)
ctx.find_correspondences(np.eye(4, dtype=np.float32), float(d["max_sq_dist"]), count=False)   # sorts under the identity
SHIFTS = ((0, 0, 0), (0.3, 0.1, 0.2), (0.5, 0.5, 0.5), (0.9, 0.8, 0.7), (1.5, 2.5, 3.5))
if len(sys.argv) > 2:
    SHIFTS = (SHIFTS[int(sys.argv[2])],)      # one case only (for a kernel trace)
for shift in SHIFTS:
    T = np.eye(4, dtype=np.float32); T[:3, 3] = np.array(shift, np.float32) * d["h"]
    ctx.find_correspondences(T, float(d["max_sq_dist"]), count=False); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.find_correspondences(T, float(d["max_sq_dist"]), count=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 5
    dq, dtl = ctx.debug_counters()
    print(f"drift {shift} cells: {dt*1e3:.3f} ms/search, deferred queries {dq}, deferred tiles {dtl}")
