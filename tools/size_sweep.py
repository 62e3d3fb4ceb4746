#!/usr/bin/env python
"""dev: ms per ICP iteration over cloud sizes (recipe pair, 20 iterations, tolerance 0; metric: p2plane / p2p)"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402
sizes = [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1e5,3e5,1e6,3e6,1e7".split(","))]
for n in sizes:
    d = syn.make_pair(n, n, with_normals=True)
    for metric in (capi.METRIC_COMBINED, capi.METRIC_POINT_TO_POINT):
        ctx = Context()
        for kv in filter(None, os.environ.get("CILHIP_OPTS", "").split(",")):      # dev: CILHIP_OPTS="group_search=0,tie_rule=0"
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(d["src"])
        p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
        p.metric, p.max_sq_dist, p.max_iter, p.conv_tol = metric, float(d["max_sq_dist"]), 20, 0.0
        ctx.icp_run(p); ctx.icp_run(p)
        ts = []
        for _ in range(5):
            ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print(f"n={n:9d} metric={'p2plane' if metric == capi.METRIC_COMBINED else 'p2p    '} {min(ts):.4f} ms/iteration (median {sorted(ts)[2]:.4f}) warm {ctx.last_warm_iterations()} forms {ctx.last_run_forms()}", flush=True)
        ctx.close()
