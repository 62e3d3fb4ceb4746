#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02aj; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -4 $O/bench_default.err; python -c "
import json
j=json.load(open('$O/bench_default.json')); r=j['roofline']
print(round(j['icp_iterations_per_sec']), j['ms_per_step'], r['frac'], r['traffic'], r['traffic_note'][:60], j['without_warm_start'], j['cpu_baseline']['value'])"
