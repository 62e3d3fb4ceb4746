#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> <cmd...>   (run on the GPU box via gpurun)
# Collects kernel-trace stats + several PMC passes (separate runs, as the guide prescribes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- "$@" > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- "$@" > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- "$@" > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- "$@" > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- "$@" > $O/sq2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/cache -- "$@" > $O/cache.log 2>&1
rocprofv3 --pmc TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $O/ta -- "$@" > $O/ta.log 2>&1
ls $O
