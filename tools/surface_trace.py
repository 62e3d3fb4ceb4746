import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from cilantro_amd import capi, synthetic as syn
from cilantro_amd.icp import Context
n = 2_000_000
rng = np.random.default_rng(1)
k = n // 2
v = rng.standard_normal((k, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
sph = 0.5 * v + np.array([0.2, 0.1, 0.0])
pl = np.stack([rng.random(n - k) * 2 - 1, rng.random(n - k) * 2 - 1, np.full(n - k, -0.5)], 1)
dst = np.concatenate([sph, pl]).astype(np.float32); nrm = np.concatenate([v, np.tile([[0, 0, 1.0]], (n - k, 1))]).astype(np.float32)
sp = float(np.sqrt(7.1 / n))
rng = np.random.default_rng(5)
noisy = dst.astype(np.float64) + rng.normal(0, 0.2 * sp, dst.shape)
Tt = np.eye(4); Tt[:3, :3] = syn.rot_xyz(0.3 * sp, -0.2 * sp, 0.25 * sp); Tt[:3, 3] = np.array([0.5, -0.3, 0.4]) * sp
Ti = np.linalg.inv(Tt)
src = (noisy @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
for r2 in (np.float32((40 * sp) ** 2), np.float32((4 * sp) ** 2)):
    ctx = Context(); ctx.set_target(dst, nrm); ctx.set_source(src)
    gi = ctx.grid_info()
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), 20, 0.0
    ctx.icp_run(p); ctx.enable_kernel_timing(True)
    ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); dt = time.perf_counter() - t0
    print("radius/spacing", float(np.sqrt(r2) / sp), "cell", gi.cell, "occ", gi.avg_occupancy, "ms/it", 1e3 * dt / 20, "forms", ctx.last_run_forms(), "warm", ctx.last_warm_iterations(), ctx.last_form_timing())
    for i, t in enumerate(ctx.last_run_trace()):
        print(i, t['form'], t['unproven'], t['listed'], t['step'] / gi.cell, t['delta'])
    ctx.close()
