cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_gpu_slab_dist.py tests/test_gpu_slab.py -x -q -m gpu > gpurun_out/r04/pytest_part.log 2>&1
echo "exit $?"; tail -8 gpurun_out/r04/pytest_part.log
