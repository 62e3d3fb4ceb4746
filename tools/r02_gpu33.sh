#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ai; mkdir -p $O; cd $R
( time timeout 2400 python -m pytest tests/ -m gpu -x -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; python -c "
import json
j=json.load(open('$O/bench_c4_1gpu.json')); r=j['roofline']
print(j['icp_iterations_per_sec'], j['ms_per_step'], r['forms_in_timed_region'], j['T_err_vs_truth_frobenius'])"
python bench.py --n 500000 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_500k.json 2> $O/bench_500k.err; python -c "
import json
j=json.load(open('$O/bench_500k.json')); r=j['roofline']
print(j['icp_iterations_per_sec'], j['ms_per_step'], r['forms_in_timed_region'])"
