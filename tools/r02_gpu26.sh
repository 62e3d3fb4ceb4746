#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02aa; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests/ -m gpu -x -q ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
