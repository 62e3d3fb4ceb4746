#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace.log 2>&1; grep -A5 "^==" $O/trace.log | cut -c1-260; grep -A24 "== frame" $O/trace.log | tail -21
WT_CASES=recipe timeout 100 python tools/warm_trace.py 1e6 20 > $O/trace_1m.log 2>&1; grep -A8 "^==" $O/trace_1m.log | cut -c1-260
WT_CASES=recipe timeout 100 python tools/warm_trace.py 1e5 20 > $O/trace_100k.log 2>&1; grep -A8 "^==" $O/trace_100k.log | cut -c1-260
WT_CASES=c4 timeout 200 python tools/warm_trace.py 1e7 20 > $O/trace_c4.log 2>&1; grep -A8 "^==" $O/trace_c4.log | cut -c1-260
timeout 900 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; tail -5 $O/test_all.log
