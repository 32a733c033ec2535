#!/bin/bash
# Developer: PMC passes for the density/force kernels.  usage: pmc.sh <tag> <config> <dist> <warm> <steps>
# (each pass under its own timeout: a counter set the hardware cannot schedule hangs rocprofv3 silently)
tag=$1; cfg=$2; dist=$3; warm=$4; steps=$5
export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
pass() { timeout -k 10 150 rocprofv3 --kernel-trace --pmc "${@:2}" --output-format csv -d ${base}_$1 -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || echo "pass $1 failed"; echo "pass $1 done"; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD
pass c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TOTAL_ACCESSES_sum
pass d TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass e TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
