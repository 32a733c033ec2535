#!/bin/bash
# HBM traffic of the step's kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": they do not fit one pass), same window as bench.py.
# usage: traffic.sh <config> <dist> <warmup> <steps>
cfg=$1; dist=$2; warm=$3; steps=$4
export TMPDIR=/tmp
base=$GRAFT_REPO_ROOT/gpurun_out/traffic_${cfg}_${dist}
rm -rf ${base}_fetch ${base}_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${base}_fetch -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d ${base}_write -- python3 tools/probe.py $cfg $dist $warm $steps > /dev/null 2>&1 || exit 1
