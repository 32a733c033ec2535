cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04p; mkdir -p $out
# the multi-rank launch path at FULL size: two gloo ranks share the one GPU (buffers through host memory: the number
# means nothing, the line and the run through step 500 do)
cp $out/bench_c3x2_one_gpu.json profiles/r04/ 2>/dev/null
( time WS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 5 --reps 1 > $out/bench_gloo_rehearsal_2ranks.json 2> $out/bench_gloo_rehearsal_2ranks.err ) 2> $out/bench_gloo_rehearsal_2ranks.time || echo "gloo rehearsal failed"
tail -3 $out/bench_gloo_rehearsal_2ranks.time
WS_BENCH_FORCE_SLAB=1 python3 bench.py --no-cpu-baseline > $out/bench_c3_cloud_slab_one_rank.json 2> $out/bench_c3_cloud_slab_one_rank.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err
echo done
