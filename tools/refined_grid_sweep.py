#!/usr/bin/env python
"""dev: option refined_occupancy_factor (how dense a REFINED grid -- surface-like target -- may stay) on the sensor frames and on synthetic
surfaces (half sphere shell, half plane), near alignment and far from it: ms per iteration of 20-iteration runs.
usage: refined_grid_sweep.py [n_synthetic ...]"""
import ctypes as C
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

sizes = [int(float(a)) for a in sys.argv[1:]] or [2_000_000]
iters = 20


def surface_cloud(n, seed=1):
    rng = np.random.default_rng(seed)
    k = n // 2
    v = rng.standard_normal((k, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    sph = 0.5 * v + np.array([0.2, 0.1, 0.0])
    pl = np.stack([rng.random(n - k) * 2 - 1, rng.random(n - k) * 2 - 1, np.full(n - k, -0.5)], 1)
    return np.concatenate([sph, pl]).astype(np.float32), np.concatenate([v, np.tile([[0, 0, 1.0]], (n - k, 1))]).astype(np.float32)


def run(name, D, N, S, r2):
    for rf in (1, 2, 3, 5):
        ctx = Context()
        ctx.set_option("refined_occupancy_factor", rf)
        ctx.set_target(D, N); ctx.set_source(S)
        gi = ctx.grid_info()
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), iters, 0.0
        ctx.icp_run(p)
        ts = []
        for _ in range(3):
            ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name:46s} factor {rf}: cell {gi.cell:.5f} occ {gi.avg_occupancy:6.2f} build {gi.build_ms:6.1f} ms  {1e3 * min(ts) / iters:.4f} ms/iteration  forms {ctx.last_run_forms()} warm {ctx.last_warm_iterations()} ncorr {int(r.last_ncorr)}", flush=True)
        ctx.close()


f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
keep = p1[:, 0] > -0.4
D, N = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep])
rng = np.random.default_rng(13)
jit = (np.float32(0.0005) * rng.uniform(-1, 1, p1.shape)).astype(np.float32)
Tm = np.eye(4); Tm[:3, :3] = syn.rot_xyz(-0.004, 0.004, -0.004); Tm[:3, 3] = [-0.003, -0.001, 0.002]
run("frame_1 vs moved+jittered frame_1", D, N, ((p1 + jit).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32), np.float32(0.01 ** 2))
run("frame_1 vs frame_2", D, N, np.ascontiguousarray(p2), np.float32(0.02 ** 2))
for n in sizes:
    dst, nrm = surface_cloud(n)
    sp = float(np.sqrt(7.1 / n))        # point spacing on the surfaces
    rng = np.random.default_rng(5)
    noisy = dst.astype(np.float64) + rng.normal(0, 0.2 * sp, dst.shape)
    for tag, scale in (("near", 1.0), ("far", 30.0)):
        Tt = np.eye(4); Tt[:3, :3] = syn.rot_xyz(0.3 * sp * scale, -0.2 * sp * scale, 0.25 * sp * scale); Tt[:3, 3] = np.array([0.5, -0.3, 0.4]) * sp * scale
        Ti = np.linalg.inv(Tt)
        src = (noisy @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        run(f"surface {n/1e6:g}M {tag} (spacing {sp:.2g}, radius {40*sp:.2g})", dst, nrm, src, np.float32((40 * sp) ** 2))
    ind, _ = surface_cloud(n, seed=9)     # an independent sampling of the same surfaces
    run(f"surface {n/1e6:g}M independent sampling", dst, nrm, ind, np.float32((40 * sp) ** 2))
