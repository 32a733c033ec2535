set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
env | grep -iE 'hsa|hip|roc|gpu_|amd' > gpurun_out/r04/env.txt || true
tools/bin/copy_probe > gpurun_out/r04/copy_probe.log 2>&1
echo copy_probe done
python3 tools/readback_timeline.py c3 sparse > gpurun_out/r04/readback_sparse.log 2>&1
echo sparse done
python3 tools/readback_timeline.py c3 settled > gpurun_out/r04/readback_settled.log 2>&1
echo settled done
rm -rf gpurun_out/r04/rb_trace
RB_FRAMES=6 RB_KINDS=registered rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/rb_trace -- python3 tools/readback_timeline.py c3 sparse > gpurun_out/r04/rb_trace.log 2>&1
ls -R gpurun_out/r04/rb_trace | head -20
