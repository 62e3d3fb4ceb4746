#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ag; mkdir -p $O; cd $R
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
export CILHIP_BENCH_FORCE_SHARDED=1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
( timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --no-extras ) > $O/sharded_weak.log 2>&1; tail -1 $O/sharded_weak.log | cut -c1-330
unset CILHIP_BENCH_FORCE_SHARDED
( time timeout 2400 python -m pytest tests/ -m gpu -x -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
