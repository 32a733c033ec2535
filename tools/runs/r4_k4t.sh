cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 500 python3 tools/k4t_check.py c2 10 400 > gpurun_out/r04/k4t_c2.log 2>&1; tail -3 gpurun_out/r04/k4t_c2.log
timeout -k 10 500 python3 tools/k4t_check.py c3 10 60 400 > gpurun_out/r04/k4t_c3.log 2>&1; tail -4 gpurun_out/r04/k4t_c3.log
