#!/usr/bin/env python
"""Dev check: non-uniform (surface-like) clouds -- adaptive grid sizing, exactness vs the oracle, speed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn
from cilantro_amd.icp import Context, SimpleCombinedMetricRigidICP3f
from oracle import oracle as orc

def surface_cloud(n, seed=1):
    rng = np.random.default_rng(seed)
    k = n // 2
    v = rng.standard_normal((k, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    sph = 0.5 * v + np.array([0.2, 0.1, 0.0])
    pl = np.stack([rng.random(n - k) * 2 - 1, rng.random(n - k) * 2 - 1, np.full(n - k, -0.5)], 1)
    pts = np.concatenate([sph, pl]).astype(np.float32)
    nrm = np.concatenate([v, np.tile([[0, 0, 1.0]], (n - k, 1))]).astype(np.float32)
    return pts, nrm

for n in (200_000,):
    dst, nrm = surface_cloud(n)
    T_true = syn.true_transform(0.01, 1.0)
    Ti = np.linalg.inv(T_true)
    rng = np.random.default_rng(5)
    src = ((dst.astype(np.float64) + rng.normal(0, 2e-4, dst.shape)) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    max_sq = np.float32(0.05 ** 2)
    icp = SimpleCombinedMetricRigidICP3f(dst, nrm, src)
    gi = icp._ctx.grid_info()
    icp.correspondenceSearchEngine().setMaxDistance(max_sq)
    icp.setMaxNumberOfIterations(30).setConvergenceTolerance(1e-6)
    t0 = time.perf_counter(); T = icp.estimate().getTransform(); dt = time.perf_counter() - t0
    print(f"n={n}: grid {gi.nx}x{gi.ny}x{gi.nz} cell={gi.cell:.4g} occ={gi.avg_occupancy:.2f} build={gi.build_ms:.1f}ms | "
          f"ICP {icp.getNumberOfPerformedIterations()} iters in {dt*1e3:.1f} ms, |T-T_true|={np.linalg.norm(T-T_true):.2e}", flush=True)
    if n <= 200_000:
        p = orc.make_params(metric=1, max_iter=30, conv_tol=1e-6, max_sq_dist=max_sq, mode=orc.MODE_MIXED)
        r = orc.icp_run(dst, nrm, src, p)
        print("   oracle iters", r["iterations"], "|T_gpu-T_oracle| = %.2e" % np.linalg.norm(T.astype(np.float64) - r["T"]))
        ctx = icp._ctx
        ctx.find_correspondences(np.eye(4), max_sq, count=False)
        gi_, gd = ctx.get_nn()
        q = orc.transform_points(np.eye(4), src)
        di, si, dv = orc.KDTree(dst).find_correspondences(q, max_sq)
        g = gi_.astype(np.int64); g[gi_ == capi.NONE_IDX] = -1
        o = np.full(len(src), -1, np.int64); o[si] = di
        print("   NN mismatches vs kd-tree:", int((g != o).sum()), "of", len(src))
        bad = np.nonzero(g != o)[0]
        bi, bd = orc.nn_brute(dst, q[bad], max_sq)
        for k, i in enumerate(bad[:10]):
            dg = np.float32(((q[i] - dst[g[i]]) ** 2).sum()) if g[i] >= 0 else None
            do = np.float32(((q[i] - dst[o[i]]) ** 2).sum()) if o[i] >= 0 else None
            print("   query", i, "gpu", g[i], gd[i], "kd", o[i], dv[np.searchsorted(si, i)] if o[i] >= 0 else None, "brute", bi[k], bd[k],
                  "same point:", bool(g[i] >= 0 and o[i] >= 0 and np.array_equal(dst[g[i]], dst[o[i]])))
