#!/bin/bash
# streaming accumulation gathering {point, normal} pairs (option pair_records): time of the accumulation pass, on / off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
for pr in 1 0; do
  timeout 200 python tools/indep_bench.py indep 1e7 12 pair_records=$pr > $O/indep_pr$pr.log 2>&1; tail -1 $O/indep_pr$pr.log | cut -c1-330
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pr -- python tools/indep_bench.py indep 1e7 12 > $O/prof_pr.log 2>&1
cp $O/prof_pr/*/*_kernel_stats.csv $O/indep_pr_kernel_stats.csv 2>/dev/null; grep "k_iter" $O/indep_pr_kernel_stats.csv | cut -c1-160
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
timeout 900 python -m pytest tests/test_gpu_loop_matches.py tests/test_gpu_parity.py -x -q -m gpu -k "not 10m and not full_size" > $O/test.log 2>&1; tail -4 $O/test.log
