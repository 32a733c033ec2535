#!/bin/bash
# The GPU jobs of a round, one name each (replaces the per-call command files of rounds 3-4).  Run through gpurun from
# the repo root:   gpurun --timeout 1200 -- 'bash tools/jobs.sh <job> [args]'
# Everything lands under gpurun_out/r05/; what is judged is copied into profiles/r05/ afterwards by hand.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export TMPDIR=/tmp WS_NO_REBUILD=1
R=r05
out=gpurun_out/$R; mkdir -p $out
job=$1; shift

case $job in
tests)
  timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; rc=$?
  echo "pytest exit $rc"; tail -5 $out/pytest_gpu.log; exit $rc
  ;;
bench)
  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err; rc=$?
  echo "bench exit $rc"; tail -c 600 $out/bench_driver_cmd.json; exit $rc
  ;;
ablate)  # ablate <warm> <lib> [<lib> ...] : one step of each tools/ab/lib<name>.so from the same state
  python3 tools/ablate.py "$@" 2>&1 | tee -a $out/ablate.log
  ;;
wg_timeline)  # wg_timeline <cfg> <state step>
  python3 tools/wg_timeline.py "$@" 2>&1 | tee -a $out/wg_timeline.log
  ;;
sched_ab)  # sched_ab <cfg> <state step> <steps>
  python3 tools/sched_ab.py "$@" 2>&1 | tee -a $out/sched_ab.log
  ;;
slab_timeline)  # the two-slab C3x2 step under the kernel trace, communication stream at the highest / the default priority
  mkdir -p $out/slab
  for pr in highest default; do
    rm -rf $out/slab/trace_$pr
    if [ $pr = default ]; then export WS_SLAB_COMM_PRIORITY=default; else unset WS_SLAB_COMM_PRIORITY; fi
    FAKE_RCCL_CHANNEL_BYTES=536870912 FAKE_RCCL_ALLGATHER_BYTES=4294967296 GPU_MAX_HW_QUEUES=24 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/slab/trace_$pr -- python3 tools/slab_timeline.py run 40 > $out/slab/run_$pr.log 2>&1 || { echo "run $pr failed"; tail -5 $out/slab/run_$pr.log; }
    python3 tools/slab_timeline.py report $(ls $out/slab/trace_$pr/*/*_kernel_trace.csv | head -1) 40 > $out/slab/timeline_two_slabs_c3x2_$pr.json
    python3 tools/step_timeline.py $(ls $out/slab/trace_$pr/*/*_kernel_trace.csv | head -1) 60 k_scan > $out/slab/step_timeline_$pr.txt 2>&1 || true
    rm -rf $out/slab/trace_$pr
  done
  unset WS_SLAB_COMM_PRIORITY
  cat $out/slab/timeline_two_slabs_c3x2_*.json
  ;;
fuzz_graph)  # fuzz_graph <cases> : violent random slab cases, captured multi-rank steps with fixed-capacity messages
  mkdir -p $out/slab
  # (full-capacity messages of the 600 000-particle cases are 36 MB each: the stand-in's staging buffers must hold them)
  FAKE_RCCL_CHANNEL_BYTES=268435456 FAKE_RCCL_ALLGATHER_BYTES=4294967296 WS_RCCL_LIBRARY=$GRAFT_REPO_ROOT/tests/libfakerccl.so GPU_MAX_HW_QUEUES=24 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 1000 python3 tools/slab_fuzz.py ${1:-100} 20261005 graph > $out/slab/slab_fuzz_graph_fixed_messages.log 2>&1
  echo "fuzz exit $?"; tail -3 $out/slab/slab_fuzz_graph_fixed_messages.log
  ;;
profiles_a)  # the judged artefacts, part 1: the driver's command (bench line, kernel trace, counters of both C3 windows)
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err || exit 1
  echo "bench driver cmd done"
  rm -rf $out/rocprof_driver_cmd
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_driver_cmd -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-north-star > $out/rocprof_driver_cmd.json 2> $out/rocprof_driver_cmd.err || exit 1
  python3 tools/window_stats.py $(ls $out/rocprof_driver_cmd/*/*_kernel_trace.csv | head -1) 5 20 > $out/rocprof_driver_cmd_windows.json
  cp $(ls $out/rocprof_driver_cmd/*/*_kernel_stats.csv | head -1) $out/rocprof_driver_cmd_kernel_stats.csv
  rm -rf $out/rocprof_driver_cmd
  echo "rocprof done"
  rm -f $out/traffic.json $out/pmc_windows.json
  for w in "5 20" "400 100"; do
    set -- $w
    tools/traffic.sh c3 cloud $1 $2 && python3 tools/traffic_report.py c3 cloud $1 $2 4194304 $out/traffic.json > /dev/null
    tools/pmc.sh ${R}c3w$1 c3 cloud $1 $2 abc > $out/pmc_c3_w$1.log 2>&1
    python3 tools/pmc_windows.py ${R}c3w$1 c3 cloud $1 $2 $out/pmc_windows.json > /dev/null
  done
  echo "c3 counters done"
  ;;
profiles_b)  # part 2: the north-star size under the counters (both windows), folded into the same traffic / pmc files
  rm -rf $out/rocprof_c4
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_c4 -- python3 bench.py --config c4 --gpus 1 --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-ieee --no-readback --no-north-star > $out/rocprof_c4.json 2> $out/rocprof_c4.err || exit 1
  python3 tools/window_stats.py $(ls $out/rocprof_c4/*/*_kernel_trace.csv | head -1) 5 20 > $out/rocprof_c4_windows.json
  cp $(ls $out/rocprof_c4/*/*_kernel_stats.csv | head -1) $out/rocprof_c4_kernel_stats.csv
  rm -rf $out/rocprof_c4
  for w in "5 20" "400 100"; do
    set -- $w
    tools/traffic.sh c4 cloud $1 $2 && python3 tools/traffic_report.py c4 cloud $1 $2 16777216 $out/traffic.json > /dev/null
    tools/pmc.sh ${R}c4w$1 c4 cloud $1 $2 abc > $out/pmc_c4_w$1.log 2>&1
    python3 tools/pmc_windows.py ${R}c4w$1 c4 cloud $1 $2 $out/pmc_windows.json > /dev/null
  done
  echo "c4 counters done"
  ;;
profiles_c)  # part 3: the other bench lines
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --graph --no-cpu-baseline --no-north-star > $out/bench_driver_cmd_graph.json 2> $out/bench_driver_cmd_graph.err
  python3 bench.py --no-north-star --no-cpu-baseline > $out/bench_c3_cloud.json 2> $out/bench_c3_cloud.err
  python3 bench.py --dist lattice --no-cpu-baseline --no-north-star > $out/bench_c3_lattice.json 2> $out/bench_c3_lattice.err
  python3 bench.py --config c2 --no-cpu-baseline > $out/bench_c2_cloud.json 2> $out/bench_c2_cloud.err
  python3 bench.py --config ref --dist lattice --no-cpu-baseline > $out/bench_ref_lattice.json 2> $out/bench_ref_lattice.err
  WS_BENCH_FORCE_SLAB=1 python3 bench.py --no-cpu-baseline --no-north-star > $out/bench_c3_cloud_slab_one_rank.json 2> $out/bench_c3_cloud_slab_one_rank.err
  echo "c3 lines done"
  python3 bench.py --config c3 --copies 2 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c3x2_one_gpu.json 2> $out/bench_c3x2_one_gpu.err
  python3 bench.py --config c4 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c4_one_gpu.json 2> $out/bench_c4_one_gpu.err
  python3 bench.py --config c5 --steps 20 --warmup 5 --reps 1 --no-cpu-baseline --no-ieee --no-readback > $out/bench_c5_one_gpu.json 2> $out/bench_c5_one_gpu.err
  echo "one-gpu lines done"
  WS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 > $out/bench_gloo_rehearsal_2ranks.json 2> $out/bench_gloo_rehearsal_2ranks.err || echo "gloo rehearsal failed"
  python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
  ;;
fuzz)  # fuzz <cases> <seed> <exact|lagged> [benign] : loopback slabs on this GPU against the single handle, bit for bit
  mkdir -p $out/slab
  timeout -k 10 1000 python3 tools/slab_fuzz.py ${1:-60} ${2:-20261006} ${3:-exact} $4 > $out/slab/slab_fuzz_${3:-exact}_seed${2:-20261006}.log 2>&1
  echo "fuzz exit $?"; tail -2 $out/slab/slab_fuzz_${3:-exact}_seed${2:-20261006}.log
  ;;
mask_words)  # the accept-mask size at C5 settled (tools/ab/libbase.so, libmw128.so, libmw256.so)
  ABLATE_CONFIG=${1:-c5} python3 tools/ablate.py 400 base mw128 mw256 2>&1 | tee $out/mask_words_${1:-c5}.log
  ;;
lib_ab)  # lib_ab <cfg> <state step> <steps> <name...> : steady-state A/B of tools/ab/lib<name>.so
  python3 tools/lib_ab.py "$@" 2>&1 | tee -a $out/lib_ab.log
  ;;
long_identity)  # long_identity <cfg> <steps>
  python3 tools/long_identity.py "$@" 2>&1 | tee -a $out/long_identity.log
  ;;
*)
  echo "unknown job $job"; exit 2
  ;;
esac
