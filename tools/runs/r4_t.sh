cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python3 tools/soak.py > $out/soak_exact.log 2>&1; echo "soak exit $?"; grep -v amdgpu.ids $out/soak_exact.log | tail -12 | cut -c1-200
