cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
ABLATE_CONFIG=c4 timeout -k 10 500 python3 tools/ablate.py 420 base mw128 mw256 > gpurun_out/r04/maskwords_c4.log 2>&1
echo "exit $?"; cat gpurun_out/r04/maskwords_c4.log
