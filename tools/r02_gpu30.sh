#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02af; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 1 > $O/devbench.log 2>&1
grep "n=" $O/devbench.log
python tools/pmc_summary.py $O k_ 2>/dev/null | grep STATS | head -8
( timeout 600 python bench.py --no-extras ) > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-330
