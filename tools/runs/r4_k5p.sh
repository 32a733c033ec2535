cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 500 python3 tools/ab_switch_check.py k5p WS_K5_PAIRS c3 10 60 400 > gpurun_out/r04/k5p_c3.log 2>&1; tail -4 gpurun_out/r04/k5p_c3.log
