set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
for churn in 0 1 2 3 5 6 7; do for pr in default high; do
  RB_CHURN=$churn WS_COPY_STREAM_PRIORITY=$pr RB_KINDS=library RB_PROFILE=1 python3 tools/readback_timeline.py c3 sparse > $out/rb3_sparse_c${churn}_${pr}.log 2>&1
done; done
echo sparse done
RB_CHURN=6 GPU_MAX_HW_QUEUES=8 WS_COPY_STREAM_PRIORITY=default RB_KINDS=library python3 tools/readback_timeline.py c3 sparse > $out/rb3_sparse_c6_default_q8.log 2>&1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_b.json 2> $out/bench_driver_cmd_b.err ) 2> $out/bench_driver_cmd_b.time
WS_COPY_STREAM_PRIORITY=default python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-north-star > $out/bench_driver_cmd_b_default_prio.json 2> $out/bench_driver_cmd_b_default_prio.err
python3 -m pytest tests/test_gpu_edge.py -x -q -m gpu -k "readback" > $out/test_readback.log 2>&1
echo bench done
