cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04p; mkdir -p $out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.log 2>&1; echo "exit $?"; tail -4 gpurun_out/r04/pytest_gpu.log
( time WS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 > $out/bench_gloo_rehearsal_2ranks.json 2> $out/bench_gloo_rehearsal_2ranks.err ) 2> $out/bench_gloo_rehearsal_2ranks.time || echo "gloo rehearsal failed"
tail -3 $out/bench_gloo_rehearsal_2ranks.time
