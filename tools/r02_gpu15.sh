#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02o; mkdir -p $O; cd $R
cat > /tmp/f6.py <<PY
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f
n = 10_000_000
d = syn.make_pair(n, n, with_normals=True)
icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"])).setPointNormalFeatureAdaptors(d["dst_n"], 0.5 * d["h"])
icp.setMaxNumberOfIterations(20).setConvergenceTolerance(0.0)
icp.estimate()
t0 = time.perf_counter(); icp.estimate(); dt = time.perf_counter() - t0
print("ms/iteration", dt * 1e3 / 20, "deferred", icp._ctx.debug_counters())
PY
for v in _clk _nog ""; do
echo "variant $v"; CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip$v.so python /tmp/f6.py 2>&1 | grep -E "phase|deferred" | tail -2
done > $O/f6var.log 2>&1
cat $O/f6var.log
