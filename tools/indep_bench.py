#!/usr/bin/env python
"""dev: the two regimes the LDS-tiled kernels serve, timed per iteration (profile with rocprofv3 --kernel-trace --stats):
  near  : the benchmark recipe with the warm-started form off (every iteration through the tiles, accumulation inside)
  indep : an independent uniform sample of the same volume as the source (matches at ~half the point spacing)"""
import ctypes as C
import os, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi, synthetic as syn  # noqa: E402
from cilantro_amd.icp import Context  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
opts = [kv.split("=") for kv in sys.argv[4:]]
d = syn.make_pair(n, n, with_normals=True)
for name in (("near", "indep") if which == "both" else (which,)):
    src = d["src"]
    if name == "indep":
        rng = np.random.default_rng(3)
        Ti = np.linalg.inv(d["T_true"].astype(np.float64))
        src = (rng.random((n, 3), dtype=np.float32).astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
    ctx = Context()
    ctx.set_option("warm_start", 0)
    for k, v in opts:
        ctx.set_option(k, float(v))
    ctx.set_target(d["dst"], d["dst_n"]); ctx.set_source(src)
    p = capi.IcpParams(); ctx._L.cilhip_icp_default_params(C.byref(p))
    p.max_sq_dist, p.max_iter, p.conv_tol = float(d["max_sq_dist"]), iters, 0.0
    ctx.icp_run(p)
    ctx.enable_kernel_timing(True)
    ctx.synchronize(); t0 = time.perf_counter(); r = ctx.icp_run(p); ctx.synchronize(); dt = time.perf_counter() - t0
    ft = ctx.last_form_timing()
    s_ms, a_ms = ctx.last_timing2()
    print(f"{name}: n={n} {1e3*dt/iters:.4f} ms/iteration  forms(one,two)={ctx.last_run_forms()} ncorr={int(r.last_ncorr)} "
          f"search-kernels {s_ms/iters:.4f} ms acc {a_ms/iters:.4f} ms  per form {{f: (ms/n, n)}} = "
          + str({f: (round(ms / k, 4), k) for f, (ms, k) in ft.items() if k}), flush=True)
    ctx.close()
