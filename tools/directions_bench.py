"""Per-iteration cost of the engine's search directions (tools only)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import CorrespondenceSearchDirection as D, SimpleCombinedMetricRigidICP3f

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
d = syn.make_pair(n, n, with_normals=True)
for direction, recip in ((D.SECOND_TO_FIRST, False), (D.FIRST_TO_SECOND, False), (D.BOTH, False), (D.BOTH, True)):
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"])).setSearchDirection(direction).setRequireReciprocality(recip)
    icp.setMaxNumberOfIterations(10).setConvergenceTolerance(0.0)
    icp.estimate()
    t0 = time.perf_counter(); icp.estimate(); dt = time.perf_counter() - t0
    T = icp.getTransform()
    print(f"n={n} {direction.name}{' reciprocal' if recip else ''}: {1e3*dt/10:.2f} ms/iteration, ncorr={icp.last_ncorr_}, |T-T_true|={np.linalg.norm(T-d['T_true']):.2e}")
