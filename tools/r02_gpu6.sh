#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab or in_tile or shard" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-2500 $O/bench_c3.json; tail -3 $O/bench_c3.err
python bench.py --config c2 > $O/bench_c2.json 2> $O/bench_c2.err; cut -c1-600 $O/bench_c2.json; tail -3 $O/bench_c2.err
python bench.py --config kmeans --steps 10 > $O/bench_kmeans.json 2> $O/bench_kmeans.err; cut -c1-1200 $O/bench_kmeans.json; tail -3 $O/bench_kmeans.err
python bench.py --config ransac --steps 8 > $O/bench_ransac.json 2> $O/bench_ransac.err; cut -c1-1200 $O/bench_ransac.json; tail -3 $O/bench_ransac.err
