#!/usr/bin/env python
"""dev: option "fused_epilogue" A/B (stage-1 reduction + epilogue in one launch against two), same process, recipe pair, 20 iterations,
tolerance 0; the transforms must be bitwise equal (same rows, same order of additions)"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402
sizes = [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1e5,1e6,1e7".split(","))]
for n in sizes:
    d = syn.make_pair(n, n, with_normals=True)
    for metric in (capi.METRIC_COMBINED, capi.METRIC_POINT_TO_POINT):
        res = {}
        for fused in (0, 1, 0, 1):
            ctx = Context(); ctx.set_option("fused_epilogue", fused); ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
            p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
            p.metric, p.max_sq_dist, p.max_iter, p.conv_tol = metric, float(d["max_sq_dist"]), 20, 0.0
            ctx.icp_run(p); ctx.icp_run(p)
            ts = []
            for _ in range(7):
                ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
            T = np.array(r.T[:], np.float32)
            res.setdefault(fused, []).append((min(ts), sorted(ts)[3], T))
            ctx.close()
        same = all(np.array_equal(a[2].view(np.uint32), res[0][0][2].view(np.uint32)) for v in res.values() for a in v)
        print(f"n={n:9d} metric={'p2plane' if metric == capi.METRIC_COMBINED else 'p2p    '} two launches {min(a[0] for a in res[0]):.4f} (median {min(a[1] for a in res[0]):.4f})  "
              f"one launch {min(a[0] for a in res[1]):.4f} (median {min(a[1] for a in res[1]):.4f}) ms/iteration  transforms bitwise equal: {same}", flush=True)
