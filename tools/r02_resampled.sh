#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02res; mkdir -p $O; cd $R
timeout 600 python tools/resampled_check.py 10000000 > $O/resampled.txt 2>&1; tail -4 $O/resampled.txt
timeout 600 python tools/warm_stress.py > $O/stress.txt 2>&1; tail -2 $O/stress.txt
python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j=json.loads(sys.stdin.read()); print(round(j['icp_iterations_per_sec']), j['ms_per_step'], j['roofline']['forms_in_timed_region'])"
