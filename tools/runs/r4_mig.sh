cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 900 python3 -m pytest tests/test_gpu_slab.py tests/test_gpu_slab_frame.py tests/test_gpu_graph.py tests/test_gpu_fake_rccl.py tests/test_gpu_slab_dist.py -x -q -m gpu > gpurun_out/r04/pytest_slab.log 2>&1; tail -3 gpurun_out/r04/pytest_slab.log
timeout -k 10 600 python3 tools/migration_peak.py c3 2 2 450 > gpurun_out/r04/migration_peak_c3x2.log 2>&1; tail -2 gpurun_out/r04/migration_peak_c3x2.log | cut -c 1-500
timeout -k 10 900 python3 tools/migration_peak.py c4 1 4 450 > gpurun_out/r04/migration_peak_c4.log 2>&1; tail -4 gpurun_out/r04/migration_peak_c4.log | cut -c 1-500
timeout -k 10 1100 python3 tools/migration_peak.py c5 1 8 450 > gpurun_out/r04/migration_peak_c5.log 2>&1; tail -8 gpurun_out/r04/migration_peak_c5.log | cut -c 1-500
