#!/bin/bash
# round 4, first GPU look at the margin records: the new route test, the loop traces, the bench line, then the whole GPU suite
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_loop_matches.py -x -q -k "margin_records" > $O/test_margin.log 2>&1; tail -15 $O/test_margin.log
timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace.log 2>&1; cat $O/trace.log | cut -c1-260
timeout 400 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
timeout 900 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; tail -15 $O/test_all.log
