cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tests/fake_rccl/selftest.py 4 100 > $out/fake_selftest_4_q8.log 2>&1
echo "selftest W=4 with 8 hardware queues: exit $?"; grep "world" $out/fake_selftest_4_q8.log
timeout -k 10 400 python3 tools/migration_peak.py c4 1 4 450 > $out/migration_peak_c4.log 2>&1; tail -4 $out/migration_peak_c4.log | cut -c 1-700
timeout -k 10 700 python3 tools/migration_peak.py c5 1 8 450 > $out/migration_peak_c5.log 2>&1; tail -8 $out/migration_peak_c5.log | cut -c 1-700
