#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02k; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "feature or tiled or clean_up" ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 600 python tools/variants_bench.py 10000000 > $O/variants.log 2>&1; cat $O/variants.log
