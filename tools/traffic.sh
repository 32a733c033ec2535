#!/bin/bash
# HBM traffic of the step's kernels, same window as bench.py, in SEPARATE --pmc passes (MI355X_MICROARCH.md "rocprofv3 PMC
# slots": the TCC block has four slots).  Reads are taken from the SIZE-RESOLVED fabric request counters of gfx950
# (round 4; tools/traffic_calib.sh): bytes = 32 n32 + 64 n64 + 128 n128.  Every request turned out to be a 128-byte line,
# and FETCH_SIZE tallies each at 64 B: 2 x FETCH_SIZE is the same number; it is kept as a cross-check.
# usage: traffic.sh <config> <dist> <warmup> <steps>
cfg=$1; dist=$2; warm=$3; steps=$4
export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/traffic_${cfg}_${dist}
rm -rf ${base}_fetch ${base}_write ${base}_rdsize
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d ${base}_rdsize -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${base}_fetch -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${base}_write -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
