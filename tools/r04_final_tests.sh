#!/bin/bash
# Round-4 evidence, part 1: the whole GPU suite (its JSON reports land in gpurun_out/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04final; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests -x -q -m gpu > $O/test_all.log 2>&1; tail -5 $O/test_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
