#!/bin/bash
# dev: the same command against several builds of the library on ONE box (CILHIP_LIB_PATH)
# usage: tools/ab_variants.sh "<tags, e.g. '' _v1 _v2>" <cmd...>   -> gpurun_out/ab_variants.txt
TAGS="$1"; shift
mkdir -p gpurun_out
{
for rep in 1 2; do
  for t in $TAGS; do
    [ "$t" = "base" ] && t=""
    echo "== libcilantro_hip$t.so (round $rep)"
    CILHIP_LIB_PATH=$PWD/cilantro_amd/lib/libcilantro_hip$t.so "$@" 2>&1 | grep -v amdgpu.ids
  done
done
} > gpurun_out/ab_variants.txt 2>&1
cat gpurun_out/ab_variants.txt
