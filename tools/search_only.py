"""Dev tool: run only the correspondence search (fixed transforms) so that PMC counters / timings of
k_search_tiled are not mixed with the rest of the loop.  usage: search_only.py [n] [reps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import Context

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
d = syn.make_pair(n, n, with_normals=False)
ctx = Context(0)
ctx.set_target(d["dst"], None)
ctx.set_source(d["src"])
ctx.set_option("tiled", 2)
T0 = np.eye(4, dtype=np.float32)
Tt = np.asarray(d["T_true"], np.float32)
for name, T in (("identity (initial misalignment)", T0), ("T_true (converged)", Tt)):
    ctx.find_correspondences(T, float(d["max_sq_dist"]), count=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.find_correspondences(T, float(d["max_sq_dist"]), count=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nf = ctx.find_correspondences(T, float(d["max_sq_dist"]))
    dq, dtl = ctx.debug_counters()
    print(f"search only, {name}: {dt*1e3:.3f} ms/search (wall incl. launch), found {nf}, deferred queries {dq} tiles {dtl}")
