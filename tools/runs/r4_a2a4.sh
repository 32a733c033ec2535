cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "exit $?"; tail -6 $out/pytest_gpu.log
