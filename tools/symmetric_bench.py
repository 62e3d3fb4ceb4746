"""dev: the symmetric point-to-plane metric (four-cloud constructor, transform_estimation.hpp:479-...) against the three-cloud one: ms per iteration, forms.
usage: symmetric_bench.py [n] [iters]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = syn.make_pair(n, n, with_normals=True)
for name, mk in (("three-cloud metric", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])),
                 ("symmetric metric (source normals = the twins')", lambda: SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"], d["dst_n"]))):
    icp = mk()
    icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"]))
    icp.setMaxNumberOfIterations(iters).setConvergenceTolerance(0.0)
    icp.estimate(); icp.estimate()
    print(f"n={n} {name}: {icp._ctx.last_timing()[0] / iters:.4f} ms/iteration, warm iterations {icp._ctx.last_warm_iterations()}, forms {icp._ctx.last_run_forms()}, "
          f"|T-T_true|={np.linalg.norm(icp.getTransform() - d['T_true']):.2e}", flush=True)
