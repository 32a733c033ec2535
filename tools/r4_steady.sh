cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=gpurun_out/r04/steady_reorder16.log
(timeout -k 10 200 python3 tools/steady.py c3 5 20 base reorder16 base reorder16 &&
timeout -k 10 200 python3 tools/steady.py c3 400 50 base reorder16 base reorder16 &&
timeout -k 10 300 python3 tools/steady.py c4 400 50 base reorder16 base reorder16) > $L 2>&1
echo "exit $?"; cat $L
