#!/bin/bash
# dev: the same commands against two builds of the library on ONE box (CILHIP_LIB_PATH): .ab/libcilantro_hip_base.so vs the tree's
# usage: tools/ab_lib.sh [sizes]      -> gpurun_out/ab_lib.txt
mkdir -p gpurun_out
S=${1:-1e5,1e6,1e7}
{
for rep in 1 2; do
  for lib in .ab/libcilantro_hip_base.so cilantro_amd/lib/libcilantro_hip.so; do
    echo "== $lib (round $rep)"
    CILHIP_LIB_PATH=$PWD/$lib python tools/size_sweep.py $S 2>&1 | grep "ms/iteration"
    CILHIP_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench it/s', round(l['icp_iterations_per_sec']), 'ms_per_step', round(l['ms_per_step'],4), 'kernel_ms_per_step', l['roofline'].get('kernel_ms_per_step'))"
  done
done
} > gpurun_out/ab_lib.txt 2>&1
cat gpurun_out/ab_lib.txt
