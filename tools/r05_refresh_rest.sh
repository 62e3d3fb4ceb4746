#!/bin/bash
# Round-5 evidence on the final build, the rest: the other configurations' lines + kernel statistics, the independent source, the regimes
# iteration by iteration, the real frames, the variants, the size sweep.  Outputs under gpurun_out/r05rest, copied to profiles/r05_* by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05rest; rm -rf $O; mkdir -p $O; cd $R
for cfg in c2 c4_1gpu; do
  C2="python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- $C2 > $O/bench_${cfg}_profiled.json 2> $O/trace_$cfg.log
  cp $O/trace_$cfg/*/*_kernel_stats.csv $O/config_${cfg}_kernel_stats.csv 2>/dev/null
done
timeout 200 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-200 $O/bench_c2.json
timeout 300 python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4_1gpu.json
WT_CASES=indep timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_indep -- python tools/warm_trace.py 10000000 20 > $O/trace_indep.log 2>&1
cp $O/trace_indep/*/*_kernel_stats.csv $O/independent_source_kernel_stats.csv 2>/dev/null
WT_CASES=recipe,indep,frames timeout 300 python tools/warm_trace.py 10000000 20 > $O/warm_trace_10m.txt 2>&1
WT_CASES=recipe,indep timeout 200 python tools/warm_trace.py 1000000 20 > $O/warm_trace_1m.txt 2>&1
WT_CASES=c4 timeout 300 python tools/warm_trace.py 10000000 20 > $O/warm_trace_c4.txt 2>&1
timeout 300 python tools/real_cloud_report.py 20 > $O/real_cloud.txt 2>&1; tail -12 $O/real_cloud.txt | cut -c1-160
timeout 150 python tools/variants_bench.py 10000000 > $O/variants.txt 2>&1; grep "n=" $O/variants.txt | cut -c1-160
timeout 100 python tools/size_sweep.py > $O/size_sweep.txt 2>&1; tail -10 $O/size_sweep.txt | cut -c1-160
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
