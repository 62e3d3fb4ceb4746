#!/bin/bash
# the core of tools/r04_final_profiles.sh after a change to the hashed sources: profiled command's line, kernel statistics, the two traffic
# passes + traffic JSON, the default line in full
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04core; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 $CMD > $O/bench_line.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err; head -4 $O/traffic.json
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; head -4 $O/bench_kernel_stats.csv | cut -c1-160
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
cp $O/traffic.json profiles/r04_traffic.json      # (so that the full line below quotes it)
timeout 500 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; cut -c1-330 $O/bench_c3_full.json
