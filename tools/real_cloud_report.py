#!/usr/bin/env python
"""The reference's sensor frames at full resolution, timed: ms per iteration and the forms the loop took (one pass / two passes / warm),
adaptive and forced, for frame_1 vs a moved + jittered copy of itself and frame_1 vs frame_2.  Default options = the reference's order of ties (tie_rule 2,
resolved on the device); the lowest-index rule (tie_rule 0) for comparison; how many tied queries a run resolved.
(correctness of every line: tests/test_gpu_loop_matches.py::test_real_sensor_frames_every_form, tests/test_gpu_tie_rule.py)
usage: real_cloud_report.py [iterations]"""
import ctypes as C
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = np.load(os.path.join(ROOT, "tests", "golden", "frames_full.npz"))
p1, n1, p2 = f["p1"], f["n1"], f["p2"]
rng = np.random.default_rng(13)
jit = (np.float32(0.0005) * rng.uniform(-1, 1, p1.shape)).astype(np.float32)
Tm = np.eye(4); Tm[:3, :3] = syn.rot_xyz(-0.004, 0.004, -0.004); Tm[:3, 3] = [-0.003, -0.001, 0.002]
src_self = ((p1 + jit).astype(np.float64) @ Tm[:3, :3].T + Tm[:3, 3]).astype(np.float32)
keep = p1[:, 0] > -0.4
D, N = np.ascontiguousarray(p1[keep]), np.ascontiguousarray(n1[keep])
cases = (("frame_1 vs moved+jittered frame_1", src_self, np.float32(0.01 * 0.01)), ("frame_1 vs frame_2", np.ascontiguousarray(p2), np.float32(0.02 * 0.02)))
forms = (("adaptive", ()), ("warm forced, per-lane start", (("warm_start", 2), ("tiled", 0), ("group_search", 0))), ("warm forced, tiled start", (("warm_start", 2), ("tiled", 2))),
         ("tiles one pass", (("warm_start", 1), ("tiled", 2), ("tile_accumulation", 2))), ("tiles two passes", (("warm_start", 0), ("tiled", 2), ("tile_accumulation", 0))),
         ("per lane", (("warm_start", 0), ("tiled", 0), ("group_search", 0))), ("adaptive, tie_rule = 0 (lowest index)", (("tie_rule", 0),)),
         ("adaptive, one lane per query only (group_search 0)", (("group_search", 0),)),
         ("8 lanes per query", (("warm_start", 0), ("tiled", 0), ("group_search", 8))),
         ("16 lanes per query", (("warm_start", 0), ("tiled", 0), ("group_search", 16))), ("32 lanes per query", (("warm_start", 0), ("tiled", 0), ("group_search", 32))))
print(f"{'registration / form':72s} {'ms/iter':>8s} {'one-pass':>8s} {'two-pass':>8s} {'warm':>5s} {'ncorr':>8s} {'step/cell (last)':>16s} {'ties resolved':>13s} {'tables ms':>9s}")
for cname, S, r2 in cases:
    for fname, opts in forms:
        ctx = Context()
        for k, v in opts:
            ctx.set_option(k, v)
        ctx.set_target(D, N); ctx.set_source(S)
        gi = ctx.grid_info()
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric, p.w_p2p, p.max_sq_dist, p.max_iter, p.conv_tol = capi.METRIC_COMBINED, 0.0, float(r2), iters, 0.0
        ctx.icp_run(p)
        ts = []
        for _ in range(3):
            ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); ts.append(time.perf_counter() - t0)
        one, two = ctx.last_run_forms(); warm = ctx.last_warm_iterations()
        tr = ctx.last_run_trace()
        seen, moved = ctx.tie_rule_stats(); ti = ctx.tie_order_info()
        print(f"{cname + ' / ' + fname:72s} {1e3 * min(ts) / iters:8.4f} {one:8d} {two:8d} {warm:5d} {int(r.last_ncorr):8d} {tr[-1]['step'] / gi.cell if tr else float('nan'):16.3g} {seen:13d} {ti['build_ms'] if ti['builds'] else 0.0:9.2f}", flush=True)
        ctx.close()
