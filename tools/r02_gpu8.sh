#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02h; mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q -k "adaptive or in_tile or end_to_end or reproduc or degenerate or 10m" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-3000 $O/bench_c3.json; tail -3 $O/bench_c3.err
