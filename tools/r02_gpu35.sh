#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ak; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests/ -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python -c "
import json
j=json.load(open('$O/bench_default.json')); r=j['roofline']
print(round(j['icp_iterations_per_sec']), j['ms_per_step'], j['source_sort_ms'], j['source_sort_ms_runs'], j['icp_estimate_ms_15iter_cold'], j['converging_run']['ms_total_incl_sort'], j['setup_ms'])"
