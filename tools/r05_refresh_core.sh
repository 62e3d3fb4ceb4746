#!/bin/bash
# Round-5 evidence, the part that follows the kernel sources' hash: the default bench command's line, kernel trace, PMC passes and the
# traffic JSON; the search directions (k_reverse_search changed); the default line in full (extras + cpu baseline).
# Run on the GPU box (gpurun); outputs under gpurun_out/r05core, copied to profiles/r05_* by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05core; rm -rf $O; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 $CMD > $O/bench_line.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1
python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
head -6 $O/traffic.json; tail -2 $O/traffic.err
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
head -6 $O/bench_kernel_stats.csv | cut -c1-160
cp $O/trace.log $O/bench_line_profiled_cmd.json 2>/dev/null
timeout 150 python tools/directions_bench.py 10000000 > $O/directions.txt 2>&1; tail -6 $O/directions.txt | cut -c1-200
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
# the traffic JSON belongs to THIS build: with it in place the default line in full (extras + cpu baseline) quotes it
cp $O/traffic.json profiles/r05_traffic.json
timeout 600 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; cut -c1-600 $O/bench_c3_full.json
