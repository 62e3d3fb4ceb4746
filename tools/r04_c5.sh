#!/bin/bash
# round 4: kernel statistics and VALU / scalar counters of the C5 configurations (KMeans3f k = 1024 on 50M points, plane RANSAC on 50M points)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c5; mkdir -p $O; cd $R
for cfg in kmeans ransac; do
  CMD="python bench.py --config $cfg --no-cpu-baseline"
  timeout 300 $CMD > $O/${cfg}_line.json 2> $O/${cfg}.err; cut -c1-700 $O/${cfg}_line.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_trace -- $CMD > $O/${cfg}_trace.log 2>&1
  cp $O/${cfg}_trace/*/*_kernel_stats.csv $O/${cfg}_kernel_stats.csv 2>/dev/null; head -6 $O/${cfg}_kernel_stats.csv | cut -c1-150
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/${cfg}_sq -- $CMD > $O/${cfg}_sq.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${cfg}_sq2 -- $CMD > $O/${cfg}_sq2.log 2>&1
done
python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
grep "k_score\|k_assign" $O/pmc_summary.txt | cut -c1-170
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
