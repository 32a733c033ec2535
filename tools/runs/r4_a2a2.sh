cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for cfg in "3 100 3000" "4 100 3000" "4 100 20000"; do
  set -- $cfg
  FAKE_RCCL_TIMEOUT_MS=$3 timeout -k 10 200 python3 tests/fake_rccl/selftest.py $1 $2 > gpurun_out/r04/fake_selftest_$1_$3.log 2>&1
  echo "selftest $cfg exit $?"; grep "world" gpurun_out/r04/fake_selftest_$1_$3.log
done
