#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
WT_CASES=indep timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace.log 2>&1; grep -A24 "^==" $O/trace.log | cut -c1-260
WT_CASES=indep CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_clk.so timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace_clk.log 2>&1; grep "warm clocks" $O/trace_clk.log | tail -3
WT_CASES=recipe CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_clk.so timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace_clk2.log 2>&1; grep "warm clocks" $O/trace_clk2.log | tail -2
timeout 600 python -m pytest tests/test_gpu_loop_matches.py -x -q -k "margin or warm_kernel_matches_index_for_index" > $O/test_loop.log 2>&1; tail -3 $O/test_loop.log
