#!/bin/bash
# size sweep of the default workload's shape (point-to-plane, Ns = Nd = n)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02an; mkdir -p $O; cd $R
for n in 100000 300000 1000000 3000000 10000000 30000000; do
python bench.py --n $n --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/b.json 2> $O/b.err
python -c "
import json
j=json.load(open('$O/b.json')); r=j['roofline']
f=r['forms_in_timed_region']
print('n=%9d  %8.0f it/s  %.4f ms/iteration  %.3e pairs/s  dominant form %d: %.4f ms = %.1f %% of the HBM roofline on algorithmic bytes; forms {form: (launches, ms)} %s' % ($n, j['icp_iterations_per_sec'], j['ms_per_step'], j['value'], r['form'], r['avg_kernel_ms'], 100*r['frac'], {k:(v['launches'], round(v['avg_kernel_ms'],4)) for k,v in f.items()}))"
done | tee $O/size_sweep.txt
