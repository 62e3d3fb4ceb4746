#!/usr/bin/env python
"""dev: the sensor frames (frame_1 vs frame_2: residuals of several cells at the default cell size) under coarser grids:
ms per iteration of the adaptive loop for cell_occupancy = 1 (default) ... 256.  usage: frames_cell_sweep.py [iters]"""
import ctypes as C
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
keep = p1[:, 0] > -0.4
from cilantro_amd import synthetic as syn  # noqa: E402
D, N = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep])
rng = np.random.default_rng(13)
jit = (np.float32(0.0005) * rng.uniform(-1, 1, p1.shape)).astype(np.float32)
Tm = np.eye(4); Tm[:3, :3] = syn.rot_xyz(-0.004, 0.004, -0.004); Tm[:3, 3] = [-0.003, -0.001, 0.002]
src_self = ((p1 + jit).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
which = os.environ.get("FCS_CASE", "frame2")
S, r2 = (np.ascontiguousarray(p2), np.float32(0.02 * 0.02)) if which == "frame2" else (src_self, np.float32(0.01 * 0.01))
print("case", which)
ref = None
for occ in (1, 2, 4, 8, 16, 32):
    for tiled in (1,):
        ctx = Context()
        ctx.set_option("cell_occupancy", occ); ctx.set_option("tiled", tiled)
        ctx.set_target(D, N); ctx.set_source(S)
        gi = ctx.grid_info()
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), iters, 0.0
        ctx.icp_run(p)
        ts = []
        for _ in range(3):
            ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        T = np.array(r.T[:], np.float32)
        if ref is None:
            ref = T
        print(f"cell_occupancy {occ:4d} tiled {tiled}: cell {gi.cell:.5f} ({np.sqrt(r2) / gi.cell:.1f} cells per radius) avg occupancy {gi.avg_occupancy:7.2f}  {1e3 * min(ts) / iters:.4f} ms/iteration  "
              f"forms {ctx.last_run_forms()} warm {ctx.last_warm_iterations()} ncorr {int(r.last_ncorr)} |T - T(default)| {float(np.abs(T - ref).max()):.2e}", flush=True)
        ctx.close()
