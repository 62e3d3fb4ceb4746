#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; mkdir -p $O; cd $R
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "weight_evaluators or accumulation_sums or combined_weights or end_to_end or symmetric or directions or in_tile or slab" ) > $O/pytest.log 2>&1
tail -30 $O/pytest.log
