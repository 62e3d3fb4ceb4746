#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02m; mkdir -p $O; cd $R
cat > /tmp/f6.py <<PY
import sys
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import SimpleCombinedMetricRigidICP3f
n = 10_000_000
d = syn.make_pair(n, n, with_normals=True)
icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"])).setPointNormalFeatureAdaptors(d["dst_n"], 0.5 * d["h"])
icp.setMaxNumberOfIterations(8).setConvergenceTolerance(0.0)
icp.estimate(); icp.estimate()
print("deferred", icp._ctx.debug_counters())
PY
CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip_clk.so python /tmp/f6.py > $O/f6clk.log 2>&1
grep -E "phase|deferred" $O/f6clk.log | tail -4
