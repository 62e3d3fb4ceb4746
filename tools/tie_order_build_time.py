#!/usr/bin/env python
"""Host time of the order tables behind option "tie_rule" (csrc/tie_order.hpp: the permutation and splits of the index the reference
builds, on the host's cores) at the benchmark's sizes, through the C ABI (cilhip_tie_order_create); the reference's own one-thread
kd-tree build of the same clouds is on the bench lines (cpu_baseline.tree_build_s: 1.8 s at 10M, 21 s at 80M).
usage: tie_order_build_time.py [sizes, default 1e6,1e7,8e7]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cilantro_amd import capi  # noqa: E402
L = capi.load()
print(f"host cores: {os.cpu_count()}")
for n in [int(float(x)) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1e6", "1e7", "8e7"])]:
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
    print(f"n = {n:9d}: order tables built in {min(ts):.3f} s (runs: {', '.join(f'{t:.3f}' for t in ts)})", flush=True)
