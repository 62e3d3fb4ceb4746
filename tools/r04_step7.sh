#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -x -q -m gpu -k "kmeans or ransac or cpp_host or model_estimation" > $O/test.log 2>&1; tail -4 $O/test.log
for cfg in kmeans ransac; do timeout 300 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('$cfg', o['ms_per_step'], o['roofline']['frac'])"; done
