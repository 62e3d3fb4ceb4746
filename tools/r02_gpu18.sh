#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02r; mkdir -p $O; cd $R
for v in _gboth _nog ""; do
echo "variant $v"; CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$v -- python tools/variants_bench.py 10000000 feat 2>&1 | grep -E "features" | tail -2
python tools/pmc_summary.py $O/trace$v k_ 2>/dev/null | grep STATS | head -4
done > $O/f6var.log 2>&1
cat $O/f6var.log
