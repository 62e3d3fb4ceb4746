#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02j; mkdir -p $O; cd $R
cat > /tmp/f2s.py <<'PY'
import sys, time
import numpy as np
sys.path.insert(0, ".")
from cilantro_amd import synthetic as syn
from cilantro_amd.icp import CorrespondenceSearchDirection as D, SimpleCombinedMetricRigidICP3f
n = 10_000_000
d = syn.make_pair(n, n, with_normals=True)
for direction in (D.FIRST_TO_SECOND, D.BOTH):
    icp = SimpleCombinedMetricRigidICP3f(d["dst"], d["dst_n"], d["src"])
    icp.correspondenceSearchEngine().setMaxDistance(float(d["max_sq_dist"])).setSearchDirection(direction)
    icp.setMaxNumberOfIterations(6).setConvergenceTolerance(0.0)
    icp.estimate(); icp.estimate()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python /tmp/f2s.py > $O/trace.log 2>&1
python tools/pmc_summary.py $O k_ 2>/dev/null | grep STATS | head -24
