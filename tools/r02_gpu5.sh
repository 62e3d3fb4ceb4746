#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02e; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 0 1 > $O/devbench.log 2>&1
cat $O/devbench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 0 > $O/trace.log 2>&1
python tools/pmc_summary.py $O k_ 2>/dev/null | grep STATS | head -12
timeout 300 python tools/drift_check.py > $O/drift.log 2>&1; tail -5 $O/drift.log
