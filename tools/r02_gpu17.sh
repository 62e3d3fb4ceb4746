#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02q; mkdir -p $O; cd $R
for v in _nog _gnp _gsn ""; do
echo "variant $v"; CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip$v.so python tools/variants_bench.py 10000000 2>&1 | grep -E "features" | tail -2
done > $O/f6var.log 2>&1
cat $O/f6var.log
