#!/bin/bash
# Regenerate the round's judged artefacts on the GPU box (run through gpurun from the repo root).
# Everything lands in gpurun_out/r03/; copy what is judged into profiles/r03/ afterwards (tools/collect_profiles.sh).
#   1. the bench line of EXACTLY the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of the same command (+ per-window summary cut from its trace)
#   3. FETCH_SIZE / WRITE_SIZE in separate --pmc passes for the driver window and the settled window
#   4. the default bench line (10..110) and the lattice / C2 lines
set -e
export TMPDIR=/tmp
out=gpurun_out/r03; mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
echo "bench driver cmd done"
rm -rf $GRAFT_REPO_ROOT/$out/rocprof_driver_cmd
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_driver_cmd -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/rocprof_driver_cmd.json 2> $out/rocprof_driver_cmd.err
trace=$(ls $out/rocprof_driver_cmd/*/*_kernel_trace.csv | head -1)
stats=$(ls $out/rocprof_driver_cmd/*/*_kernel_stats.csv | head -1)
python3 tools/window_stats.py $trace 5 20 > $out/rocprof_driver_cmd_windows.json
cp $stats $out/rocprof_driver_cmd_kernel_stats.csv
echo "rocprof done"
tools/traffic.sh c3 cloud 5 20 && python3 tools/traffic_report.py c3 cloud 5 20 4194304 $out/traffic.json
tools/traffic.sh c3 cloud 400 100 && python3 tools/traffic_report.py c3 cloud 400 100 4194304 $out/traffic.json
echo "traffic done"
python3 bench.py > $out/bench_c3_cloud.json 2> $out/bench_c3_cloud.err
python3 bench.py --dist lattice --no-cpu-baseline > $out/bench_c3_lattice.json 2> $out/bench_c3_lattice.err
python3 bench.py --config c2 --no-cpu-baseline > $out/bench_c2_cloud.json 2> $out/bench_c2_cloud.err
# the slab path with one rank through the native RCCL transport, same window as bench_c3_cloud (overhead of the slab step)
WS_BENCH_FORCE_SLAB=1 python3 bench.py --no-cpu-baseline > $out/bench_c3_cloud_slab_one_rank.json 2> $out/bench_c3_cloud_slab_one_rank.err
# copy the first window's traffic into place and print the driver line once more, now with roofline.traffic filled
mkdir -p profiles/r03 && cp $out/traffic.json profiles/r03/traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
# SQ / TA counters of the bench windows (roofline.secondary: VALU issue, texture addresser)
tools/pmc.sh r03w5 c3 cloud 5 20 abc > $out/pmc_w5.log 2>&1; python3 tools/pmc_windows.py r03w5 c3 cloud 5 20 $out/pmc_windows.json
tools/pmc.sh r03w400 c3 cloud 400 100 abc > $out/pmc_w400.log 2>&1; python3 tools/pmc_windows.py r03w400 c3 cloud 400 100 $out/pmc_windows.json
cp $out/pmc_windows.json profiles/r03/pmc_windows.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
echo done
