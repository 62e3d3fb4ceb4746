#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; cd $R
timeout 600 python tools/variants_bench.py 10000000 > $O/variants.log 2>&1; cat $O/variants.log
CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip_clk.so timeout 600 python tools/variants_bench.py 10000000 2>&1 | grep -E "phase" | tail -6 > $O/clk.log; cat $O/clk.log
