cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_slab.py tests/test_gpu_slab_exact.py tests/test_gpu_slab_frame.py tests/test_gpu_fake_rccl.py tests/test_gpu_host.py -x -q -m gpu > $out/pytest_part.log 2>&1
echo "exit $?"; tail -5 $out/pytest_part.log
