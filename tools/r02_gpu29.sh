#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ad; mkdir -p $O; cd $R
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 0 2 1 > $O/devbench.log 2>&1
timeout 600 python tools/devbench.py --n 10000000 --modes 0 --tiled 1 --tacc 1 --warm 2 --metric p2p >> $O/devbench.log 2>&1
grep "n=" $O/devbench.log
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "warm or shard or slab or adaptive or end_to_end" ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
export CILHIP_BENCH_FORCE_SHARDED=1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
( timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --no-extras ) > $O/sharded_weak.log 2>&1; tail -1 $O/sharded_weak.log | cut -c1-330
unset CILHIP_BENCH_FORCE_SHARDED
( timeout 600 python bench.py --no-extras ) > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-330
