#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "feature or tiled or clean_up or adaptive or in_tile" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python tools/variants_bench.py 10000000 feat > $O/variants.log 2>&1; grep features $O/variants.log
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r02s/trace/runc/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
for name in ("k_search_tiled","k_search_deferred","k_iter"):
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if name in r["Kernel_Name"]]
    print(name,[round(x) for x in d])
PY
timeout 300 python tools/drift_check.py > $O/drift.log 2>&1; tail -14 $O/drift.log
