#!/bin/bash
# Developer: build kernel variants of the library for in-session A/B runs (tools/ab/*.so; select with WSFLUID_LIBRARY).
# usage: ab_build.sh name "-DFLAG=.. -DFLAG=.." [srcdir]
set -e
root=$(cd $(dirname $0)/..; pwd)
name=$1; flags=$2; src=${3:-$root}
mkdir -p $root/tools/ab
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-unused-result $flags \
  -I $src/include -I $src/water-sandbox_amd/csrc -o $root/tools/ab/lib$name.so \
  $src/water-sandbox_amd/csrc/ws_kernels.hip $src/water-sandbox_amd/csrc/ws_api.cpp $src/water-sandbox_amd/csrc/ws_rccl.cpp $src/water-sandbox_amd/csrc/ws_local.cpp -ldl
echo built tools/ab/lib$name.so
