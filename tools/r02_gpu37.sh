#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02am; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests/ -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json
j=json.load(open('$O/bench.json')); print(round(j['icp_iterations_per_sec']), j['ms_per_step'], j['setup_ms'])"
