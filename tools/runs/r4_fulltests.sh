cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.log 2>&1
echo "exit $?"; tail -5 gpurun_out/r04/pytest_gpu.log
