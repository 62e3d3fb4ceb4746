"""dev: the affine classes' device-resident loop at size n -- iterations per kernel form and their average kernel times, against the
host-driven loop (option affine_device_loop = 0).  usage: affine_forms.py [n] [iterations]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import SimpleCombinedMetricAffineICP3f, SimplePointToPointMetricAffineICP3f

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = syn.make_pair(n, n, with_normals=True)
for name, mk in (("affine combined", lambda: SimpleCombinedMetricAffineICP3f(d["dst"], d["dst_n"], d["src"])),
                 ("affine point-to-point", lambda: SimplePointToPointMetricAffineICP3f(d["dst"], d["src"]))):
    for loop in (1, 0):
        icp = mk()
        icp._ctx.set_option("affine_device_loop", loop)
        icp._ctx.set_option("kernel_timing", 1)
        icp._ctx.set_option("kernel_timing_stride", 4)
        icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
        icp.setMaxNumberOfIterations(iters).setConvergenceTolerance(0.0)
        icp.estimate()
        icp.estimate()
        loop_ms, _, _ = icp._ctx.last_timing()
        print(f"n={n} {name} device_loop={loop}: {loop_ms / iters:.4f} ms/iteration (events around the loop), warm iterations {icp._ctx.last_warm_iterations()}, "
              f"forms {icp._ctx.last_form_timing()}, ncorr={icp.last_ncorr_}, |T-T_true|={np.linalg.norm(icp.getTransform() - d['T_true']):.2e}", flush=True)
