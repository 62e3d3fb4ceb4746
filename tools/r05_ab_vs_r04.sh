# dev: this build against the round-4 build (a git worktree of the round-4 commit under .r04tmp, built there) on the same box
set -e
if [ -d .r04tmp ]; then (cd .r04tmp && python tools/size_sweep.py 1e5,1e6,1e7 2>/dev/null && python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('r04 bench', round(d['ms_per_step'],4), {k:(v['launches'],round(v['avg_kernel_ms'],4)) for k,v in d['roofline']['forms_in_timed_region'].items()}, 'nowarm', round(d['without_warm_start']['ms_per_step'],4), 'indep', round(d['independent_source']['ms_per_step'],4), 'c15', round(d['icp_estimate_ms_15iter_cold'],3), 'conv', round(d['converging_run']['ms_per_iteration'],4))
"); fi
echo ---- new
python tools/size_sweep.py 1e5,1e6,1e7 2>/dev/null
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('new bench', round(d['ms_per_step'],4), {k:(v['launches'],round(v['avg_kernel_ms'],4)) for k,v in d['roofline']['forms_in_timed_region'].items()}, 'nowarm', round(d['without_warm_start']['ms_per_step'],4), 'indep', round(d['independent_source']['ms_per_step'],4), 'c15', round(d['icp_estimate_ms_15iter_cold'],3), 'conv', round(d['converging_run']['ms_per_iteration'],4))
"
