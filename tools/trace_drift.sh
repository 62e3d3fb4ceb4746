#!/bin/bash
# usage (on the GPU box): tools/trace_drift.sh <case> ...   -- kernel-trace stats of one drift_check case each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for c in "$@"; do
  O=$R/gpurun_out/drift$c; mkdir -p $O
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tools/drift_check.py 10000000 $c > $O/log.txt 2>&1
  tail -1 $O/log.txt
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -7 "$f" | cut -d, -f1-5; fi
done
