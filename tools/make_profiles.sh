#!/bin/bash
# Regenerate the round's judged artefacts on the GPU box (run through gpurun from the repo root).
# Everything lands in gpurun_out/r04p/; copy what is judged into profiles/r04/ afterwards.
#   1. the bench line of EXACTLY the driver's command (python3 bench.py --gpus 1 --steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of the same command (+ per-window summary cut from its trace)
#   3. size-resolved read requests / FETCH_SIZE / WRITE_SIZE in separate --pmc passes for both windows
#   4. SQ / TA counter passes of both windows
#   5. the other bench lines (defaults, lattice, C2, one-rank slab, graph, C4 / C5 / C3x2 on one GPU, gloo rehearsal)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04p; mkdir -p $out profiles/r04
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
echo "bench driver cmd done"
rm -rf $GRAFT_REPO_ROOT/$out/rocprof_driver_cmd
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_driver_cmd -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-north-star > $out/rocprof_driver_cmd.json 2> $out/rocprof_driver_cmd.err
trace=$(ls $out/rocprof_driver_cmd/*/*_kernel_trace.csv | head -1)
stats=$(ls $out/rocprof_driver_cmd/*/*_kernel_stats.csv | head -1)
python3 tools/window_stats.py $trace 5 20 > $out/rocprof_driver_cmd_windows.json
cp $stats $out/rocprof_driver_cmd_kernel_stats.csv
echo "rocprof done"
tools/traffic.sh c3 cloud 5 20 && python3 tools/traffic_report.py c3 cloud 5 20 4194304 $out/traffic.json > /dev/null
tools/traffic.sh c3 cloud 400 100 && python3 tools/traffic_report.py c3 cloud 400 100 4194304 $out/traffic.json > /dev/null
cp $out/traffic.json profiles/r04/traffic.json
echo "traffic done"
tools/pmc.sh r04w5 c3 cloud 5 20 abc > $out/pmc_w5.log 2>&1; python3 tools/pmc_windows.py r04w5 c3 cloud 5 20 $out/pmc_windows.json > /dev/null
tools/pmc.sh r04w400 c3 cloud 400 100 abc > $out/pmc_w400.log 2>&1; python3 tools/pmc_windows.py r04w400 c3 cloud 400 100 $out/pmc_windows.json > /dev/null
cp $out/pmc_windows.json profiles/r04/pmc_windows.json
echo "pmc done"
# the driver line once more, now with roofline.traffic / secondary / step_traffic filled from this round's counter passes
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-north-star > $out/bench_driver_cmd_graph.json 2> $out/bench_driver_cmd_graph.err
python3 bench.py --no-north-star > $out/bench_c3_cloud.json 2> $out/bench_c3_cloud.err
python3 bench.py --dist lattice --no-cpu-baseline > $out/bench_c3_lattice.json 2> $out/bench_c3_lattice.err
python3 bench.py --config c2 --no-cpu-baseline > $out/bench_c2_cloud.json 2> $out/bench_c2_cloud.err
WS_BENCH_FORCE_SLAB=1 python3 bench.py --no-cpu-baseline > $out/bench_c3_cloud_slab_one_rank.json 2> $out/bench_c3_cloud_slab_one_rank.err
echo "c3 lines done"
# the like-for-like yard-sticks of the multi-GPU lines: the N-GPU configurations on ONE GPU, in the driver's window
python3 bench.py --config c3 --copies 2 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c3x2_one_gpu.json 2> $out/bench_c3x2_one_gpu.err
python3 bench.py --config c4 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c4_one_gpu.json 2> $out/bench_c4_one_gpu.err
python3 bench.py --config c5 --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c5_one_gpu.json 2> $out/bench_c5_one_gpu.err
echo "one-gpu lines done"
WS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 > $out/bench_gloo_rehearsal_2ranks.json 2> $out/bench_gloo_rehearsal_2ranks.err || echo "gloo rehearsal failed"
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1
echo done
