#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02i; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "directions or affine or post_filters or symmetric or frame1" ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
timeout 600 python tools/directions_bench.py 10000000 > $O/directions.log 2>&1; cat $O/directions.log
