#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02ah; mkdir -p $O; cd $R
( timeout 600 python bench.py --no-cpu-baseline ) > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print(j['icp_iterations_per_sec'], j['ms_per_step'], r['forms_in_timed_region'], j['icp_estimate_ms_15iter_cold'], j['converging_run'])"
export CILHIP_BENCH_FORCE_SHARDED=1
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517"
( timeout 600 $TR bench.py --gpus 1 --steps 20 --warmup 3 --no-extras ) > $O/sharded_weak.log 2>&1; tail -1 $O/sharded_weak.log | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print(j['icp_iterations_per_sec'], j['ms_per_step'], r['forms_in_timed_region'])"
unset CILHIP_BENCH_FORCE_SHARDED
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "warm or adaptive or shard or slab or end_to_end" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
