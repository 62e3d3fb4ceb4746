#!/bin/bash
# round 4: loop-match tests, loop traces, kernel statistics of the default bench command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_loop_matches.py -x -q > $O/test_loop.log 2>&1; tail -5 $O/test_loop.log
timeout 300 python tools/warm_trace.py 1e7 20 > $O/trace.log 2>&1; grep -A4 "^==" $O/trace.log | cut -c1-260
CMD="python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace_bench.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null; head -12 $O/bench_kernel_stats.csv | cut -c1-160
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,os
o=json.load(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r04b/bench.json"))
print({k:o[k] for k in ("icp_iterations_per_sec","ms_per_step","icp_estimate_ms_15iter_cold")})
print("without_warm", o["without_warm_start"]["ms_per_step"], "converging", o["converging_run"])
print("indep", {k:v for k,v in o["independent_source"].items() if k!="roofline"})
print("forms", o["roofline"]["forms_in_timed_region"], "cold", o["roofline_cold"]["avg_kernel_ms"] if o.get("roofline_cold") else None)
PY
