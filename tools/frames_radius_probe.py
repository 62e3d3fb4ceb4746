#!/usr/bin/env python
"""dev: what the per-lane search of the sensor frames spends its time on: one search (cilhip_find_correspondences) of frame_2 against frame_1
under the identity and under the converged transform, for shrinking radii (a query with nothing inside the radius walks every shell inside it)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402
f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
keep = p1[:, 0] > -0.4
D, N, S = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep]), np.ascontiguousarray(p2)
ctx = Context(); ctx.set_target(D, N); ctx.set_source(S)
gi = ctx.grid_info()
p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(np.float32(0.02 ** 2)), 40, 0.0
r = ctx.icp_run(p)
Tc = np.array(r.T[:], np.float32).reshape(4, 4).T
for name, T in (("identity", np.eye(4, dtype=np.float32)), ("converged", Tc)):
    for rad in (0.02, 0.01, 0.005, 0.0025):
        r2 = float(np.float32(rad * rad))
        ctx.find_correspondences(T, r2, count=False)
        ts = []
        for _ in range(5):
            ctx.synchronize(); t0 = time.perf_counter(); n = ctx.find_correspondences(T, r2); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:10s} radius {rad:.4f} = {rad / gi.cell:4.1f} cells: {1e3 * min(ts):.4f} ms per search, {n} of {len(S)} matched", flush=True)
ctx.close()
