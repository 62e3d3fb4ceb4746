#!/bin/bash
# k_warm block by block (dev build with clocks): where do the 20 us between the average block's lifetime and the kernel's duration go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
WT_CASES=recipe CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_clk.so timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace_clk.log 2>&1; grep "warm clocks\|warm stamps" $O/trace_clk.log | tail -8
