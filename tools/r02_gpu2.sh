#!/bin/bash
# round-2 GPU call 2: the in-tile accumulation -- parity first, then A/B timing and a kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02b; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "in_tile or tiled or end_to_end or degenerate or reproducible or shard" ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 0 1 > $O/devbench.log 2>&1
cat $O/devbench.log
timeout 600 python tools/devbench.py --n 1000000 --modes 0 --tiled 1 --tacc 0 1 --metric p2p >> $O/devbench.log 2>&1
tail -2 $O/devbench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_trace.log 2>&1
python tools/pmc_summary.py $O k_ 2>/dev/null | grep STATS | head -12
