#!/bin/bash
# dev: the default bench command under variant builds of the library (CILHIP_LIB_PATH); prints ms/step and the per-form kernel times
for lib in "$@"; do
  L=""; [ "$lib" != "default" ] && L="cilantro_amd/lib/libcilantro_hip_$lib.so"
  CILHIP_LIB_PATH=$L timeout 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', 'ms/step', round(d['ms_per_step'],4), 'forms', r.get('forms_in_timed_region'))"
done
