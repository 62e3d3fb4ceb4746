#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02bench; mkdir -p $O; cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json
j=json.load(open('$O/bench_default.json')); r=j['roofline']
print(round(j['icp_iterations_per_sec']), j['ms_per_step'], r['form'], r['frac'], r['traffic'], j['cpu_baseline']['value'])"
