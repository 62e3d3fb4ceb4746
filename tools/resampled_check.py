#!/usr/bin/env python
"""A source that is NOT the target's points plus small noise: an independent uniform sample of the same volume, so that nearest
distances are a good fraction of the point spacing and the nearest-other-point table settles only part of the queries.  The
adaptive loop must notice (listed queries > 1/8) and stay with the tiled kernels; forced warm-started iterations (warm_start =
2) show what it avoids.  Results are the same in all three."""
import ctypes as C
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
d = syn.make_pair(n, n, with_normals=True)
rng = np.random.default_rng(3)
src = rng.random((n, 3), dtype=np.float32)
Ti = np.linalg.inv(d["T_true"].astype(np.float64))
src = (src.astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
# second case (argv[2] = noise in units of the point spacing): the target's points plus LARGE noise -- matches near enough for the
# tiles' octant stage to prove them, too far for the table to settle most of them: the case the listed-query count is for
if len(sys.argv) > 2:
    d = syn.make_pair(n, n, with_normals=True, noise=float(sys.argv[2]))
    src = d["src"]
ref = None
for warm in (0, 1, 2):
    ctx = Context(); ctx.set_option("warm_start", warm)
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(src)
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.max_sq_dist, p.max_iter, p.conv_tol = float(d["max_sq_dist"]), 12, 0.0
    ctx.icp_run(p)
    ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); dt = time.perf_counter() - t0
    T = np.array(r.T[:], np.float64)
    if ref is None:
        ref = (T, int(r.last_ncorr))
    print(f"n={n} warm_start={warm}: {1e3*dt/12:.3f} ms/iteration, warm iterations {ctx.last_warm_iterations()} of 12, ncorr {int(r.last_ncorr)} (ref {ref[1]}), "
          f"max |T - T(warm_start=0)| = {np.abs(T - ref[0]).max():.2e}", flush=True)
    assert int(r.last_ncorr) == ref[1] and np.abs(T - ref[0]).max() <= 5e-7
    ctx.close()
