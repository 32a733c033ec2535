#!/bin/bash
# Developer: PMC passes for the density/force kernels.  usage: pmc.sh <tag> <config> <dist> <warm> <steps> [passes]
# (each pass under its own timeout: a counter set the hardware cannot schedule hangs rocprofv3 silently)
tag=$1; cfg=$2; dist=$3; warm=$4; steps=$5; passes=${6:-abc}
export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
pass() { [[ $passes == *$1* ]] || return 0; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${@:2}" --output-format csv -d ${base}_$1 -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || echo "pass $1 failed"; echo "pass $1 done"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
pass c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
pass l SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL
