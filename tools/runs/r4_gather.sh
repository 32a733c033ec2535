cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout -k 10 120 tools/bin/gather_probe > gpurun_out/r04/gather_probe.log 2>&1
echo "exit $?"; cat gpurun_out/r04/gather_probe.log
