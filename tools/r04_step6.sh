#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -x -q -m gpu -k "tie_count or cpp_host or end_to_end or solve" > $O/test.log 2>&1; tail -5 $O/test.log
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace_bench.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; grep "k_solve\|k_reduce\|k_warm" $O/bench_kernel_stats.csv | cut -c1-130
CMD="python bench.py --config c2 --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -- $CMD > $O/trace_bench2.log 2>&1
cp $O/trace2/*/*_kernel_stats.csv $O/c2_kernel_stats.csv 2>/dev/null; grep "k_solve\|k_reduce\|k_warm" $O/c2_kernel_stats.csv | cut -c1-130
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
timeout 100 python bench.py --config c2 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-400
timeout 100 python tools/size_sweep.py 2>&1 | tail -8 | cut -c1-200
