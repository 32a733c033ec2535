#!/bin/bash
# Developer: build a kernel variant of the library for in-session A/B runs (tools/ab/lib<name>.so; select with
# WSFLUID_LIBRARY or tools/ablate.py).  The shipped sources carry no experiment switches: an experiment is a PATCH under
# tools/ab/patches/ (ablations whose results are deliberately wrong, instrumentation, alternatives measured and not
# adopted -- HISTORY.md), applied to a scratch copy of the sources here.
# usage: ab_build.sh <name> "<-D flags for the tuning macros, may be empty>" [patch name ...]
#   e.g. ab_build.sh nomask "" ablate_mask_store        ab_build.sh p128 "-DND_P=128"
set -e
root=$(cd $(dirname $0)/..; pwd)
name=$1; flags=$2; shift 2 || true
work=$(mktemp -d /tmp/wsab.XXXXXX)
mkdir -p $work/water-sandbox_amd $work/include $root/tools/ab
cp -r $root/water-sandbox_amd/csrc $work/water-sandbox_amd/csrc
cp $root/include/wsfluid.h $work/include/
for p in "$@"; do
  patch -s -p1 -d $work < $root/tools/ab/patches/$p.patch
  echo "applied $p"
done
src=$work/water-sandbox_amd/csrc
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-unused-result -DWS_DEV_HOOKS $flags \
  -I $work/include -I $src -o $root/tools/ab/lib$name.so \
  $src/ws_kernels.hip $src/ws_api.cpp $src/ws_rccl.cpp $src/ws_local.cpp -ldl
rm -rf $work
echo built tools/ab/lib$name.so
