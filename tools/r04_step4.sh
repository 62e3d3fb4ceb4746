#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
echo "== default"; timeout 200 python tools/indep_bench.py both 1e7 12 2>&1 | grep "n=" | cut -c1-330
echo "== edge 9, occupancy 2.5"; CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_e9.so timeout 200 python tools/indep_bench.py both 1e7 12 cell_occupancy=2.5 2>&1 | grep "n=\|rror" | cut -c1-330
echo "== edge 9, occupancy 2"; CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_e9.so timeout 200 python tools/indep_bench.py both 1e7 12 cell_occupancy=2 2>&1 | grep "n=\|rror" | cut -c1-330
echo "== edge 8, occupancy 3.4"; CILHIP_LIB_PATH=cilantro_amd/lib/libcilantro_hip_e8.so timeout 200 python tools/indep_bench.py both 1e7 12 cell_occupancy=3.4 2>&1 | grep "n=\|rror" | cut -c1-330
WT_CASES=frames timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace_frames.log 2>&1; cat $O/trace_frames.log | cut -c1-200
timeout 900 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; tail -5 $O/test_all.log
