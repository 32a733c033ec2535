cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
export WS_RCCL_LIBRARY=$PWD/tests/libfakerccl.so GPU_MAX_HW_QUEUES=24
WSFLUID_LIBRARY=$PWD/tools/ab/libsplitold.so timeout -k 10 200 python3 tests/fake_rccl_thin_worker.py 2>$out/thin_old.err | grep "^{" | cut -c1-700
