#!/bin/bash
# Round-3 evidence: bench line + kernel trace + PMC passes + traffic JSON of the default bench command, then the bench lines of
# the other configs.  Run on the GPU box (gpurun); outputs under gpurun_out/r03final, copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03final; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 $CMD > $O/bench_line.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1
python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
head -30 $O/traffic.json; tail -2 $O/traffic.err
grep STATS $O/pmc_summary.txt | head -12
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
timeout 400 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; cut -c1-600 $O/bench_c3_full.json
timeout 200 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
timeout 300 python bench.py --config kmeans > $O/bench_kmeans.json 2> $O/bench_kmeans.err; cut -c1-300 $O/bench_kmeans.json
timeout 300 python bench.py --config ransac > $O/bench_ransac.json 2> $O/bench_ransac.err; cut -c1-300 $O/bench_ransac.json
timeout 300 python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; cut -c1-300 $O/bench_c4_1gpu.json
timeout 200 tools/bin/read_bw_probe > $O/read_bw_probe.txt 2>&1; tail -3 $O/read_bw_probe.txt
timeout 150 python tools/variants_bench.py 10000000 > $O/variants.txt 2>&1; grep "n=" $O/variants.txt | cut -c1-200
timeout 150 python tools/directions_bench.py 10000000 > $O/directions.txt 2>&1; tail -6 $O/directions.txt | cut -c1-200
timeout 100 python tools/size_sweep.py > $O/size_sweep.txt 2>&1; tail -8 $O/size_sweep.txt | cut -c1-200
