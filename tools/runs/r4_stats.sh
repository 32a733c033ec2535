cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=gpurun_out/r04/state_stats_c3_c4.log
(timeout -k 10 200 python3 tools/state_stats.py c3 cloud 420 && timeout -k 10 400 python3 tools/state_stats.py c4 cloud 420) > $L 2>&1
echo "exit $?"; cat $L
