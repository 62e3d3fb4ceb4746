#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02al; mkdir -p $O; cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
python -c "
import json
for f in ('bench_default','bench_c4_1gpu'):
    j=json.load(open('$O/%s.json'%f)); r=j['roofline']
    print(f, round(j['icp_iterations_per_sec']), j['ms_per_step'], r['form'], r['frac'], r['traffic'], j.get('cpu_baseline',{}).get('value'))"
