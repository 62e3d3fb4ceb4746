#!/usr/bin/env python
"""dev: the adaptive loop iteration by iteration (form, unproven, listed, step / cell, kernel time) on the regimes the
warm-started form has to serve: the recipe, an independently sampled source, the reference's real sensor frames.
usage: warm_trace.py [n] [iters] [key=value ...]"""
import ctypes as C
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
opts = [kv.split("=") for kv in sys.argv[3:]]


def run(name, D, N, S, r2, metric=capi.METRIC_COMBINED):
    ctx = Context()
    for k, v in opts:
        ctx.set_option(k, float(v))
    ctx.set_target(D, N); ctx.set_source(S)
    gi = ctx.grid_info()
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.metric, p.max_sq_dist, p.max_iter, p.conv_tol = metric, float(r2), iters, 0.0
    ctx.icp_run(p)
    ctx.enable_kernel_timing(True)
    ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); dt = time.perf_counter() - t0
    ft = ctx.last_form_timing()
    tr = ctx.last_run_trace()
    print(f"== {name}: ns={len(S)} nd={len(D)} cell={gi.cell:.5g} occ={gi.avg_occupancy:.2f} {1e3*dt/iters:.4f} ms/iteration ncorr={int(r.last_ncorr)} warm={ctx.last_warm_iterations()} "
          f"forms(one,two)={ctx.last_run_forms()} per form (ms, n)=" + str({f: (round(ms / k, 4), k) for f, (ms, k) in ft.items() if k}), flush=True)
    print("   it form unproven listed step/cell delta")
    for i, t in enumerate(tr):
        print(f"   {i:2d} {t['form']} {t['unproven']:9d} {t['listed']:9d} {t['step']/gi.cell:10.4g} {t['delta']:10.3g}")
    ctx.close()


which = os.environ.get("WT_CASES", "recipe,indep,frames").split(",")
if "recipe" in which or "indep" in which:
    d = syn.make_pair(n, n, with_normals=True, noise=float(os.environ.get("WT_NOISE", "0.05")))
    if "recipe" in which:
        run("recipe", d["dst"], d["dst_n"], d["src"], d["max_sq_dist"])
        run("recipe p2p", d["dst"], None, d["src"], d["max_sq_dist"], capi.METRIC_POINT_TO_POINT)
    if "indep" in which:
        rng = np.random.default_rng(3)
        Ti = np.linalg.inv(d["T_true"].astype(np.float64))
        si = (rng.random((n, 3), dtype=np.float32).astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        run("independent", d["dst"], d["dst_n"], si, d["max_sq_dist"])
if "frames" in which:
    f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
    p1, n1, p2 = f["p1"], f["n1"], f["p2"]
    keep = p1[:, 0] > -0.4
    D, N = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep])
    run("frame_1 vs frame_2", D, N, np.ascontiguousarray(p2), np.float32(0.02 * 0.02))
if "c4" in which:
    d = syn.make_pair(80_000_000, 10_000_000, with_normals=True, src_stride=8)
    run("c4 shape: 10M vs 80M", d["dst"], d["dst_n"], d["src"], d["max_sq_dist"])
