cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 1000 python3 -m pytest tests/test_gpu_slab_exact.py -x -q -m gpu > gpurun_out/r04/pytest_part.log 2>&1
echo "exit $?"; tail -25 gpurun_out/r04/pytest_part.log
