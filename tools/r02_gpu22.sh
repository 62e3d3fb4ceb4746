#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02v; mkdir -p $O; cd $R
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 0 2 1 > $O/devbench.log 2>&1
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 0 2 --metric p2p >> $O/devbench.log 2>&1
cat $O/devbench.log
