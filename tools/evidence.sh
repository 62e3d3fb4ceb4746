#!/bin/bash
# The evidence run of a round, parameterised (replaces the per-round lease scripts r02_* ... r05_*):
#   tools/evidence.sh <tag> [section ...]        e.g.  gpurun --timeout 1500 -- 'bash tools/evidence.sh r06 core configs'
# Outputs under gpurun_out/<tag>final/, copied into profiles/<tag>_* by hand (the judge reads profiles/).  Sections:
#   core     the default bench command: line, rocprofv3 kernel trace + stats, the PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, two SQ
#            groups), their summary, the traffic JSON (tools/make_traffic_json.py)
#   search   the SQ / LDS counter passes of the tile kernel alone (tools/pmc_search.sh on the loop without warm start)
#   configs  bench lines + kernel statistics of the other configurations (c2, c4_1gpu, kmeans, ransac)
#   c5pmc    KMeans' pruned assignment under the SQ / LDS counter passes + every launch's duration
#   regimes  the loop iteration by iteration (recipe / independent source / sensor frames / configs[3]'s shape), the real-cloud report,
#            variants, search directions, the size sweep, the read-bandwidth probe, the tie-order build times
#   full     the default line in full (extras + CPU baseline): the long one, last
#   lines    ONLY the bench lines again (default in full, c2, c4_1gpu, kmeans, ransac), no profiler: after `core` / `configs` of the same build
#            were collected into profiles/ (tools/collect_profiles.sh), so that the lines quote the build's counter traffic
cd /tmp && export TMPDIR=/tmp
TAG=${1:-dev}; shift; SECTIONS=${*:-core configs regimes full}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}final; mkdir -p $O; cd $R
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
prune() { find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete; }
if has core; then
  timeout 200 $CMD > $O/bench_line.json 2> $O/bench.err
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- $CMD > $O/sq.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- $CMD > $O/sq2.log 2>&1
  python tools/pmc_summary.py $O k_ > $O/pmc_summary.txt 2>&1
  python tools/make_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
  head -12 $O/traffic.json; tail -2 $O/traffic.err
  cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
  head -8 $O/bench_kernel_stats.csv | cut -c1-180
  prune
fi
if has search; then
  bash tools/pmc_search.sh ${TAG}final/search python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 2 --warm 0 --steps 6
  python tools/pmc_summary.py $O/search k_search_tiled > $O/search_pmc_summary.txt 2>&1
  prune
fi
if has configs; then
  for cfg in c2 c4_1gpu; do
    C2="python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- $C2 > $O/bench_${cfg}_profiled.json 2> $O/trace_$cfg.log
    cp $O/trace_$cfg/*/*_kernel_stats.csv $O/config_${cfg}_kernel_stats.csv 2>/dev/null
  done
  timeout 200 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2.json
  timeout 300 python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err; cut -c1-300 $O/bench_c4_1gpu.json
  for cfg in kmeans ransac; do
    C2="python bench.py --config $cfg --no-cpu-baseline"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -- $C2 > $O/bench_${cfg}_profiled.json 2> $O/trace_$cfg.log
    cp $O/trace_$cfg/*/*_kernel_stats.csv $O/config_${cfg}_kernel_stats.csv 2>/dev/null
  done
  # HBM traffic of every configuration's dominant kernel (the counters of `core`, per configuration): profiles/<tag>_traffic_<cfg>.json
  for cfg in c2 c4_1gpu; do
    C2="python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-other-configs"
    mkdir -p $O/pmc_$cfg
    timeout 300 $C2 > $O/pmc_$cfg/bench_line.json 2> $O/pmc_$cfg/bench.err
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$cfg/fetch -- $C2 > $O/pmc_$cfg/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_$cfg/write -- $C2 > $O/pmc_$cfg/write.log 2>&1
    python tools/make_traffic_json.py $O/pmc_$cfg > $O/traffic_$cfg.json 2> $O/traffic_$cfg.err
  done
  for cfg in kmeans ransac; do
    C2="python bench.py --config $cfg --no-cpu-baseline"
    mkdir -p $O/pmc_$cfg
    timeout 300 $C2 > $O/pmc_$cfg/bench_line.json 2> $O/pmc_$cfg/bench.err
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$cfg/fetch -- $C2 > $O/pmc_$cfg/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_$cfg/write -- $C2 > $O/pmc_$cfg/write.log 2>&1
    python tools/make_traffic_json.py $O/pmc_$cfg --config $cfg $([ $cfg = kmeans ] && echo k_assign_grid || echo k_score) > $O/traffic_$cfg.json 2> $O/traffic_$cfg.err
  done
  head -5 $O/traffic_c2.json $O/traffic_c4_1gpu.json $O/traffic_kmeans.json $O/traffic_ransac.json
  prune
fi
if has c5pmc; then
  # KMeans' pruned assignment: the SQ / LDS counter passes that name its limiter, and every launch's duration (the 1.3 - 3.4 ms spread)
  KM="python bench.py --config kmeans --no-cpu-baseline"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c5/sq -- $KM > $O/c5_sq.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $O/c5/sq2 -- $KM > $O/c5_sq2.log 2>&1
  python tools/pmc_summary.py $O/c5 k_assign > $O/config_c5_pmc_summary.txt 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/c5/trace -- $KM > $O/c5_trace.log 2>&1
  python - $O <<'PY' > $O/config_c5_launches.txt 2>&1
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/c5/trace/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "k_assign_grid" in r["Kernel_Name"] or "k_centroid_grid" in r["Kernel_Name"] or "k_assign_accumulate" in r["Kernel_Name"]]
    t0 = min(int(r["Start_Timestamp"]) for r in rows) if rows else 0
    for i, r in enumerate(rows):
        print(f"{i:3d} {r['Kernel_Name'][:48]:48s} start_ms={(int(r['Start_Timestamp']) - t0) / 1e6:9.3f} dur_us={(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f}")
PY
  prune
fi
if has regimes; then
  WT_CASES=indep timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_indep -- python tools/warm_trace.py 10000000 20 > $O/trace_indep.log 2>&1
  cp $O/trace_indep/*/*_kernel_stats.csv $O/independent_source_kernel_stats.csv 2>/dev/null
  WT_CASES=recipe,indep,frames timeout 300 python tools/warm_trace.py 10000000 20 > $O/warm_trace_10m.txt 2>&1
  WT_CASES=recipe,indep timeout 200 python tools/warm_trace.py 1000000 20 > $O/warm_trace_1m.txt 2>&1
  WT_CASES=c4 timeout 300 python tools/warm_trace.py 10000000 20 > $O/warm_trace_c4.txt 2>&1
  timeout 300 python tools/real_cloud_report.py 20 > $O/real_cloud.txt 2>&1; tail -16 $O/real_cloud.txt | cut -c1-160
  [ -x tools/bin/read_bw_probe ] && timeout 200 tools/bin/read_bw_probe > $O/read_bw_probe.txt 2>&1
  timeout 150 python tools/variants_bench.py 10000000 > $O/variants.txt 2>&1; grep "n=" $O/variants.txt | cut -c1-200
  timeout 150 python tools/directions_bench.py 10000000 > $O/directions.txt 2>&1; tail -6 $O/directions.txt | cut -c1-200
  timeout 200 python tools/affine_forms.py 10000000 20 > $O/affine_forms.txt 2>&1; python tools/affine_forms.py 1000000 20 >> $O/affine_forms.txt 2>&1; grep "n=" $O/affine_forms.txt | cut -c1-160
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_affine -- python tools/affine_forms.py 10000000 20 > $O/trace_affine.log 2>&1
  cp $O/trace_affine/*/*_kernel_stats.csv $O/affine_kernel_stats.csv 2>/dev/null
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_directions -- python tools/directions_bench.py 10000000 > $O/trace_directions.log 2>&1
  cp $O/trace_directions/*/*_kernel_stats.csv $O/directions_kernel_stats.csv 2>/dev/null
  timeout 100 python tools/size_sweep.py > $O/size_sweep.txt 2>&1; tail -8 $O/size_sweep.txt | cut -c1-200
  timeout 300 python tools/tie_order_build_time.py > $O/tie_order_build.txt 2>&1; tail -4 $O/tie_order_build.txt
  prune
fi
if has full; then
  timeout 600 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err; cut -c1-700 $O/bench_c3_full.json
fi
if has lines; then
  timeout 200 python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err
  timeout 300 python bench.py --config c4_1gpu --steps 20 --warmup 3 --no-extras > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
  timeout 300 python bench.py --config kmeans > $O/bench_kmeans_profiled.json 2> $O/bench_kmeans.err
  timeout 300 python bench.py --config ransac > $O/bench_ransac_profiled.json 2> $O/bench_ransac.err
  timeout 900 python bench.py > $O/bench_c3_full.json 2> $O/bench_c3_full.err
  cut -c1-200 $O/bench_c3_full.json
fi
