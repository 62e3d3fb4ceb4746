#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02u; mkdir -p $O; cd $R
export CILHIP_BENCH_FORCE_SHARDED=1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
( timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --no-extras ) > $O/sharded_weak.log 2>&1; tail -2 $O/sharded_weak.log
( timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --no-extras --scaling strong ) > $O/sharded_strong.log 2>&1; tail -2 $O/sharded_strong.log
( timeout 900 $TR bench.py --gpus 1 --steps 10 --warmup 2 --no-extras --config c4 ) > $O/sharded_c4.log 2>&1; tail -2 $O/sharded_c4.log
unset CILHIP_BENCH_FORCE_SHARDED
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
