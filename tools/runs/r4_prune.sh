cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 400 python3 tools/ab_compare.py base prune c3 12 70 420 > $out/prune_compare.log 2>&1; tail -3 $out/prune_compare.log | cut -c 1-300
for w in 10 60 400; do timeout -k 10 300 python3 tools/ablate.py $w base prune base prune >> $out/prune_ablate.log 2>&1; done; tail -12 $out/prune_ablate.log | cut -c 1-220
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_kat.py tests/test_gpu_small_radius.py tests/test_gpu_overflow.py tests/test_gpu_slab.py -x -q -m gpu > $out/pytest_prune.log 2>&1; tail -3 $out/pytest_prune.log
