cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 600 python3 -m pytest tests/test_gpu_slab.py -x -q -m gpu -k "far_route" > gpurun_out/r04/pytest_part.log 2>&1
echo "exit $?"; tail -12 gpurun_out/r04/pytest_part.log
