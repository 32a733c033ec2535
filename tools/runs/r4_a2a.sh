cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
WS_RCCL_LIBRARY=$PWD/tests/libfakerccl.so timeout -k 10 200 python3 tests/fake_rccl/selftest.py 4 120 > gpurun_out/r04/fake_selftest.log 2>&1
echo "selftest exit $?"; tail -5 gpurun_out/r04/fake_selftest.log
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu.log 2>&1
echo "exit $?"; tail -25 gpurun_out/r04/pytest_gpu.log
