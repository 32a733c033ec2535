set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
python3 -m pytest tests/test_gpu_edge.py -x -q -m gpu -k "readback" > $out/test_readback.log 2>&1
echo tests done
for prof in 0 1; do for copy in kernel runtime; do
  RB_PROFILE=$prof WS_READBACK_COPY=$copy RB_KINDS=registered,library python3 tools/readback_timeline.py c3 sparse > $out/rb2_sparse_p${prof}_${copy}.log 2>&1
done; done
echo sparse done
RB_KINDS=library python3 tools/readback_timeline.py c3 settled > $out/rb2_settled_p0_kernel.log 2>&1
RB_PROFILE=1 WS_READBACK_COPY=runtime RB_KINDS=library python3 tools/readback_timeline.py c3 settled > $out/rb2_settled_p1_runtime.log 2>&1
echo settled done
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd_a.json 2> $out/bench_driver_cmd_a.err ) 2> $out/bench_driver_cmd_a.time
echo bench done
rm -rf $out/rb_trace2
RB_FRAMES=6 RB_KINDS=library rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rb_trace2 -- python3 tools/readback_timeline.py c3 sparse > $out/rb_trace2.log 2>&1
echo trace done
