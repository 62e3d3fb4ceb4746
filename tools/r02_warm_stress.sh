#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02stress; mkdir -p $O; cd $R
( time timeout 1500 python tools/warm_stress.py ) > $O/stress.log 2>&1; tail -12 $O/stress.log
