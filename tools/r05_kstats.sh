#!/bin/bash
# dev: rocprofv3 kernel statistics of the profiled bench command; $1 = label, CILHIP_LIB_PATH selects a variant library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k_$1; rm -rf $O; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline $2"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
echo "== $1"
python - <<PY
import csv
for r in csv.reader(open('$O/bench_kernel_stats.csv')):
    if any(k in r[0] for k in ('k_search','k_warm','k_reduce','k_solve','k_tile','k_iter','k_init')):
        print(r[0][:64].ljust(64), r[1].rjust(4), ('%.1f' % (float(r[3])/1000)).rjust(8), ('%.1f' % (float(r[5])/1000)).rjust(8), ('%.1f' % (float(r[6])/1000)).rjust(8))
PY
