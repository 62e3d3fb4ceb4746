#!/bin/bash
# profiles of the round-2 bench command: kernel trace + PMC passes, traffic JSON
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02g; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
$CMD > $O/bench_line.json 2> $O/bench.err
bash tools/pmc.sh r02g $CMD > $O/pmc.log 2>&1
python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
cat $O/traffic.json; tail -2 $O/traffic.err
grep STATS $O/pmc_summary.txt | head -12
grep -E "k_warm<2, 2>" $O/pmc_summary.txt | head -40
