#!/bin/bash
# early warm decision (second iteration) + warm_extra_fraction 0.0625: margin routes test, then the regimes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_loop_matches.py -x -q -m gpu -k "margin_records or real_sensor" > $O/test.log 2>&1; tail -5 $O/test.log
WT_CASES=recipe,indep,frames timeout 200 python tools/warm_trace.py 10000000 20 > $O/trace.log 2>&1; grep "==" $O/trace.log | cut -c1-260
WT_CASES=recipe,indep timeout 200 python tools/warm_trace.py 1000000 20 > $O/trace_1m.log 2>&1; grep "==" $O/trace_1m.log | cut -c1-260
