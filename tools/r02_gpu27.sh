#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ab; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "warm_started" ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
