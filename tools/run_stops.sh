cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for v in _stop0 _stop1 _stop2 _noov _nofull _nofullnoov ""; do
  O=$R/gpurun_out/stops/v$v; mkdir -p $O
  CILHIP_LIB_PATH=$R/cilantro_amd/lib/libcilantro_hip$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O -- python tools/search_only.py 1e7 3 > $O/log.txt 2>&1
  echo "variant [$v]"; tail -2 $O/log.txt
  python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_search_tiled" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"  {c:24s} n={len(v)} first={v[0]:.4g} last={v[-1]:.4g}")
PY
done
