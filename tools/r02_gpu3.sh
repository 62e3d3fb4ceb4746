#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "in_tile or clean_up or reproducible" ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 0 1 > $O/devbench.log 2>&1
cat $O/devbench.log
CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip_clk.so timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 0 > $O/devbench_clk.log 2>&1
grep -E "phase|n=" $O/devbench_clk.log | tail -12
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 > $O/trace.log 2>&1
python tools/pmc_summary.py $O k_ 2>/dev/null | grep STATS | head -8
