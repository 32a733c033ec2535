#!/bin/bash
# Developer: PMC passes for the density/force kernels.  usage: pmc.sh <tag> <config> <dist> <warm> <steps>
tag=$1; cfg=$2; dist=$3; warm=$4; steps=$5
export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d ${base}_a -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD --output-format csv -d ${base}_b -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum --output-format csv -d ${base}_c -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || echo "pass c failed"
