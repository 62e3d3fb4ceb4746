#!/usr/bin/env python
"""Build time of the order tables behind option "tie_rule" (csrc/tie_build.hip: the permutation and splits of the index the reference
builds, level by level on the device) at the benchmark's sizes: (a) a context's own build from its grid (cilhip_build_tie_order:
everything on the device), (b) through cilhip_tie_order_create (host cloud in, host tables out: + upload and download).  The
reference's own one-thread kd-tree build of the same clouds is on the bench lines (cpu_baseline.tree_build_s: 1.8 s at 10M, 21 s at
80M); round 5 built these tables on the host's 256 cores in 0.008 / 0.27 / 2.6 s (120k / 10M / 80M).
usage: tie_order_build_time.py [sizes, default 1.2e5,1e6,1e7,8e7]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi  # noqa: E402
L = capi.load()
print(f"host cores: {os.cpu_count()}")
for n in [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1.2e5", "1e6", "1e7", "8e7"])]:
    rng = np.random.default_rng(0)
    D = rng.random((n, 3), dtype=np.float32)
    ts = []
    for _ in range(2):
        h = C.c_void_p()
        t0 = time.perf_counter()
        rc = L.cilhip_tie_order_create(D.ctypes.data, n, C.byref(h))
        ts.append(time.perf_counter() - t0)
        assert rc == 0
        L.cilhip_tie_order_destroy(h)
    from cilantro_amd.icp import Context
    ctx = Context(0)
    ctx.set_target(D, None)
    tb = []
    for _ in range(3):
        ctx.set_option("tie_rule", 2)
        ctx.build_tie_order()
        tb.append(ctx.tie_order_info()["build_ms"])
        ctx.set_target(D, None)      # (drops the tables)
    ctx.close()
    print(f"n = {n:9d}: context build (device only) {min(tb):8.2f} ms (runs: {', '.join(f'{t:.2f}' for t in tb)});  cilhip_tie_order_create (host in / host out) "
          f"{min(ts) * 1e3:8.1f} ms (runs: {', '.join(f'{t * 1e3:.1f}' for t in ts)})", flush=True)
