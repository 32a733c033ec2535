cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
bash tools/r4_k5p.sh
bash tools/traffic_calib.sh > $out/calib.log 2>&1; tail -3 $out/calib.log
export WSFLUID_LIBRARY=$GRAFT_REPO_ROOT/tools/ab/libk5p.so
WS_K4_CELLS=1 tools/pmc.sh k4t c3 cloud 400 5 ab > $out/pmc_k4t.log 2>&1; python3 tools/pmc_windows.py k4t c3 cloud 400 5 $out/pmc_experiments_k4t.json > /dev/null
WS_K5_PAIRS=1 tools/pmc.sh k5p c3 cloud 400 5 abc > $out/pmc_k5p.log 2>&1; python3 tools/pmc_windows.py k5p c3 cloud 400 5 $out/pmc_experiments_k5p.json > /dev/null
tools/pmc.sh base c3 cloud 400 5 abc > $out/pmc_base.log 2>&1; python3 tools/pmc_windows.py base c3 cloud 400 5 $out/pmc_experiments_base.json > /dev/null
echo pmc done
