#!/bin/bash
# dev: per-kernel average times (rocprofv3 --kernel-trace --stats) of one command under several builds of the library
# usage: tools/ab_kstats.sh "<tags: base _v1 ...>" <kernel-substring> <cmd...>   -> gpurun_out/ab_kstats.txt
TAGS="$1"; FILT="$2"; shift; shift
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $R
{
for t in $TAGS; do
  [ "$t" = "base" ] && t=""
  rm -rf /tmp/kst; CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip$t.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- "$@" > /tmp/kst.log 2>&1
  echo "== libcilantro_hip$t.so"
  python - "$FILT" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/kst/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Name"]:
            print("  %-90s calls=%s avg_us=%.1f min_us=%.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
} > gpurun_out/ab_kstats.txt 2>&1
cat gpurun_out/ab_kstats.txt
