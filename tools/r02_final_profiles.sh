#!/bin/bash
# Round-2 evidence: bench line + kernel trace + PMC passes + traffic JSON of the default bench command, then the bench
# lines of the other configs and the variant / direction / drift tables.  Run on the GPU box (gpurun); outputs under
# gpurun_out/r02final, copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02final; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
$CMD > $O/bench_line.json 2> $O/bench.err
bash tools/pmc.sh r02final $CMD > $O/pmc.log 2>&1
python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
cat $O/traffic.json | head -30; tail -2 $O/traffic.err
grep STATS $O/pmc_summary.txt | head -12
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +1M -delete
python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; cut -c1-400 $O/bench_c3_full.json
python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
python bench.py --config kmeans > $O/bench_kmeans.json 2> $O/bench_kmeans.err; cut -c1-300 $O/bench_kmeans.json
python bench.py --config ransac > $O/bench_ransac.json 2> $O/bench_ransac.err; cut -c1-300 $O/bench_ransac.json
python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; cut -c1-300 $O/bench_c4_1gpu.json
timeout 600 python tools/variants_bench.py 10000000 > $O/variants.txt 2>&1; grep "n=" $O/variants.txt
timeout 600 python tools/directions_bench.py 10000000 > $O/directions.txt 2>&1; tail -8 $O/directions.txt
timeout 300 python tools/drift_check.py > $O/drift.txt 2>&1; tail -6 $O/drift.txt
