cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
A=${STEADY_A:-base}; B=${STEADY_B:-w8}
L=gpurun_out/r04/steady_${B}.log
(timeout -k 10 200 python3 tools/steady.py c3 5 20 $A $B $A $B &&
timeout -k 10 200 python3 tools/steady.py c3 60 20 $A $B $A $B &&
timeout -k 10 200 python3 tools/steady.py c3 400 50 $A $B $A $B) > $L 2>&1
echo "exit $?"; cut -c1-260 $L
