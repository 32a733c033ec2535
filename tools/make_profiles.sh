#!/bin/bash
# Regenerate the round's judged artefacts on the GPU box (run through gpurun from the repo root):
#   bench lines (cloud + lattice), rocprofv3 --kernel-trace --stats of the SAME bench command,
#   FETCH_SIZE / WRITE_SIZE passes, variant A/B table.  Outputs land in gpurun_out/r01/.
set -e
export TMPDIR=/tmp
out=gpurun_out/r01; mkdir -p $out
python bench.py > $out/bench_c3_cloud.json 2> $out/bench_c3_cloud.err
python bench.py --dist lattice --no-cpu-baseline > $out/bench_c3_lattice.json 2> $out/bench_c3_lattice.err
python bench.py --config c2 --no-cpu-baseline > $out/bench_c2_cloud.json 2> $out/bench_c2_cloud.err
rm -rf $GRAFT_REPO_ROOT/$out/rocprof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_bench -- python3 bench.py --no-cpu-baseline > $out/rocprof_bench.json 2> $out/rocprof_bench.err
tools/traffic.sh c3 cloud 10 100
: > $out/variants.log
for w in 10 60 200; do
  WS_VARIANT=simple python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out/variants.log
  WS_VARIANT=simple WS_IEEE=1 python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out/variants.log
  python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out/variants.log
  WS_IEEE=1 python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out/variants.log
done
echo done
