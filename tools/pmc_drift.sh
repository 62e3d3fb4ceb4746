#!/bin/bash
# usage (on the GPU box): tools/pmc_drift.sh <tag> <case>   -- instruction-mix counters of one drift_check case
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/pmcdrift_$1; mkdir -p $O
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python tools/drift_check.py 10000000 $2 > $O/a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- python tools/drift_check.py 10000000 $2 > $O/b.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcdrift_$1 k_search_tiled 2>&1 | grep -v STATS
