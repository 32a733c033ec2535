cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_slab.py tests/test_gpu_slab_frame.py tests/test_gpu_fake_rccl.py tests/test_gpu_graph.py -x -q -m gpu > $out/pytest_slab.log 2>&1; tail -3 $out/pytest_slab.log
rm -rf $GRAFT_REPO_ROOT/$out/rocprof_c4_4slabs
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/rocprof_c4_4slabs -- python3 tools/migration_peak.py c4 1 4 450 > $out/migration_peak_c4.log 2>&1; tail -4 $out/migration_peak_c4.log | cut -c 1-420
cp $(ls $out/rocprof_c4_4slabs/*/*_kernel_stats.csv | head -1) $out/rocprof_c4_4slabs_kernel_stats.csv; head -14 $out/rocprof_c4_4slabs_kernel_stats.csv | cut -c 1-200
timeout -k 10 1100 python3 tools/migration_peak.py c5 1 8 450 > $out/migration_peak_c5.log 2>&1; tail -8 $out/migration_peak_c5.log | cut -c 1-420
timeout -k 10 600 python3 tools/migration_peak.py c3 2 2 450 > $out/migration_peak_c3x2.log 2>&1; tail -2 $out/migration_peak_c3x2.log | cut -c 1-420
