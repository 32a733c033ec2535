set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_fake_rccl.py -x -q -m gpu > $out/test_fake_rccl.log 2>&1 || { tail -40 $out/test_fake_rccl.log; exit 1; }
tail -3 $out/test_fake_rccl.log
