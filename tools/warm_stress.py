#!/usr/bin/env python
"""Stress of the warm-started iterations: loops with option warm_start = 1 / 2 against warm_start = 0 on many (seeded) cloud
pairs -- uniform and surface-like, several sizes, metrics, Gauss-Newton step counts, start offsets, fixed iteration counts and
tolerance-gated runs.  Same iterations, same correspondence counts, transforms equal to the order of the f64 additions."""
import ctypes as C
import sys, os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402


def bumpy(n, seed):
    rng = np.random.default_rng(seed)
    u = rng.random((n, 2)).astype(np.float32)
    z = (0.15 * np.sin(6.0 * u[:, 0]) * np.cos(4.0 * u[:, 1]) + 0.5).astype(np.float32)
    p = np.stack([u[:, 0], u[:, 1], z], 1)
    nrm = np.stack([-0.9 * np.cos(6.0 * u[:, 0]) * np.cos(4.0 * u[:, 1]), 0.6 * np.sin(6.0 * u[:, 0]) * np.sin(4.0 * u[:, 1]), np.ones(n)], 1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(p), np.ascontiguousarray(nrm)


def run(dst, dst_n, src, r2, metric, w_p2p, steps, max_iter, tol, warm, tiled):
    ctx = Context()
    ctx.set_option("warm_start", warm)
    if tiled is not None:
        ctx.set_option("tiled", tiled)
    ctx.set_target(dst, dst_n); ctx.set_source(src)
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol, p.max_opt_iter = metric, w_p2p, float(r2), max_iter, tol, steps
    r = ctx.icp_run(p)
    out = (np.array(r.T[:], np.float64), int(r.iterations), int(r.last_ncorr), ctx.last_warm_iterations())
    ctx.close()
    return out


bad = 0
cases = 0
for seed, n, pert in ((1, 70_000, 0.3), (2, 200_000, 0.6), (3, 1_500_000, 0.3), (4, 1_500_000, 0.9), (5, 400_000, 0.2)):
    d = syn.make_pair(n, perturb=pert)
    clouds = [("uniform", d["dst"], d["dst_n"], d["src"], d["max_sq_dist"])]
    if seed in (2, 5):
        bp, bn = bumpy(n, seed)
        rng = np.random.default_rng(seed)
        T = syn.true_transform(0.01, 0.5)
        bs = ((bp + rng.normal(scale=2e-4, size=bp.shape).astype(np.float32)) @ np.linalg.inv(T)[:3, :3].T.astype(np.float32) + np.linalg.inv(T)[:3, 3].astype(np.float32)).astype(np.float32)
        clouds.append(("surface", bp, bn, np.ascontiguousarray(bs), 4e-4))
    for name, D, N, S, r2 in clouds:
        for metric, w_p2p, steps in ((capi.METRIC_COMBINED, 0.0, 1), (capi.METRIC_COMBINED, 0.2, 2), (capi.METRIC_POINT_TO_POINT, 0.0, 1)):
            for max_iter, tol in ((12, 0.0), (40, 1e-6)):
                for tiled in (None, 2):
                    ref = run(D, N, S, r2, metric, w_p2p, steps, max_iter, tol, 0, tiled)
                    for warm in (1, 2):
                        got = run(D, N, S, r2, metric, w_p2p, steps, max_iter, tol, warm, tiled)
                        cases += 1
                        dT = np.abs(got[0] - ref[0]).max()
                        ok = got[1] == ref[1] and got[2] == ref[2] and dT <= 5e-7
                        if not ok:
                            bad += 1
                            print("MISMATCH", name, n, pert, metric, w_p2p, steps, max_iter, tol, tiled, warm, "iters", got[1], ref[1], "ncorr", got[2], ref[2], "dT", dT)
    print(f"seed {seed} n {n} perturb {pert}: done, warm iterations of the last run {got[3]}", flush=True)
print(f"{cases} comparisons, {bad} mismatches")
sys.exit(1 if bad else 0)
