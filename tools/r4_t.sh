cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 1000 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "c5_in_eight" > gpurun_out/r04/pytest_new.log 2>&1; tail -15 gpurun_out/r04/pytest_new.log | cut -c 1-300
