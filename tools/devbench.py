#!/usr/bin/env python
"""Developer A/B bench: per-kernel timings of the ICP loop for several options (not the driver's bench)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[1_000_000, 10_000_000])
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--metric", default="p2plane")
ap.add_argument("--modes", type=int, nargs="+", default=[1, 0])
ap.add_argument("--occ", type=float, nargs="+", default=[1.0])
ap.add_argument("--tiled", type=int, nargs="+", default=[1])
ap.add_argument("--tacc", type=int, nargs="+", default=[1], help="tile_accumulation option values to compare")
ap.add_argument("--warm", type=int, nargs="+", default=[1], help="warm_start option values to compare")
a = ap.parse_args()

for n in a.n:
    d = syn.make_pair(n, n, with_normals=True)
    for fused, occ, tiled, tacc, warm in [(f, o, t, ta, w) for f in a.modes for o in a.occ for t in a.tiled for ta in a.tacc for w in a.warm]:
        ctx = Context(0)
        ctx.set_option("cell_occupancy", occ)
        ctx.set_target(d["dst"], d["dst_n"] if a.metric == "p2plane" else None)
        ctx.set_source(d["src"])
        ctx.set_option("fused", fused)
        ctx.set_option("tiled", 2 if tiled else 0)
        ctx.set_option("tile_accumulation", tacc)
        ctx.set_option("warm_start", warm)
        p = capi.IcpParams()
        ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric = capi.METRIC_COMBINED if a.metric == "p2plane" else capi.METRIC_POINT_TO_POINT
        p.conv_tol = 0.0
        p.max_sq_dist = float(d["max_sq_dist"])
        p.max_iter = 3
        ctx.icp_run(p)
        p.max_iter = a.steps
        ctx.enable_kernel_timing(False)
        t0 = time.perf_counter(); r = ctx.icp_run(p); dt = time.perf_counter() - t0
        loop_ms, _, _ = ctx.last_timing()
        ctx.enable_kernel_timing(True)
        r = ctx.icp_run(p)
        loop2, sk, nl = ctx.last_timing()
        s_ms, a_ms = ctx.last_timing2()
        dq, dtl = ctx.debug_counters()
        gi = ctx.grid_info()
        T = np.array(r.T[:], np.float32).reshape(4, 4).T
        print(f"n={n} fused={fused} occ={occ} tiled={tiled} tile_acc={tacc} warm={warm} (warm iterations {ctx.last_warm_iterations()}) wall/iter={dt*1e3/a.steps:.3f}ms loop(ev)/iter={loop_ms/a.steps:.3f}ms "
              f"[timed: loop/iter={loop2/a.steps:.3f} search(or fused)/iter={s_ms/max(nl,1):.3f} acc/iter={a_ms/max(nl,1):.3f}] "
              f"it/s={a.steps/dt:.1f} err_true={np.linalg.norm(T-d['T_true']):.2e} ncorr={r.last_ncorr} deferred_queries={dq} deferred_tiles={dtl} grid={gi.nx}x{gi.ny}x{gi.nz}", flush=True)
        ctx.close()
