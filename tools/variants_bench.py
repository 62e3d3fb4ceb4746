"""Per-iteration cost of the affine ICP instances and of the 6-D point+normal feature search (tools only)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import (SimpleCombinedMetricRigidICP3f, SimpleCombinedMetricAffineICP3f,
                              SimplePointToPointMetricAffineICP3f)

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
d = syn.make_pair(n, n, with_normals=True)
iters = 10
only_feat = len(sys.argv) > 2 and sys.argv[2] == 'feat'
for name, mk in () if only_feat else (("rigid combined", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])),
                 ("affine combined", lambda: SimpleCombinedMetricAffineICP3f(d["dst"], d["dst_n"], d["src"])),
                 ("affine point-to-point", lambda: SimplePointToPointMetricAffineICP3f(d["dst"], d["src"]))):
    icp = mk()
    icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
    icp.setMaxNumberOfIterations(iters).setConvergenceTolerance(0.0)
    icp.estimate()
    t0 = time.perf_counter(); icp.estimate(); dt = time.perf_counter() - t0
    loop_ms, _, nl = icp._ctx.last_timing()
    print(f"n={n} {name}: {1e3*dt/iters:.3f} ms/iteration by wall clock of estimate() (includes the per-call source sort), "
          f"{loop_ms/iters:.3f} ms/iteration by events around the loop; ncorr={icp.last_ncorr_}, |T-T_true|={np.linalg.norm(icp.getTransform()-d['T_true']):.2e}")
# feature search: source normals = the target's (roughly right for a near-aligned pair)
icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"])).setPointNormalFeatureAdaptors(d["dst_n"], 0.5 * d["h"])
icp.setMaxNumberOfIterations(iters).setConvergenceTolerance(0.0)
icp.estimate()
t0 = time.perf_counter(); icp.estimate(); dt = time.perf_counter() - t0
loop_ms, _, nl = icp._ctx.last_timing()
print(f"n={n} rigid combined, point+normal features (w = h/2): {1e3*dt/iters:.3f} ms/iteration by wall clock of estimate(), "
      f"{loop_ms/iters:.3f} ms/iteration by events around the loop; ncorr={icp.last_ncorr_} deferred={icp._ctx.debug_counters()}")
