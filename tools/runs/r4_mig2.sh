cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 1100 python3 tools/migration_peak.py c5 1 8 450 > gpurun_out/r04/migration_peak_c5.log 2>&1; tail -8 gpurun_out/r04/migration_peak_c5.log | cut -c 1-520
