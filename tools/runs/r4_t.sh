cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 1000 python3 tools/slab_fuzz.py 1000 12345 exact > $out/slab_fuzz_exact_1000.log 2>&1; echo "exact 1000: exit $? $(tail -1 $out/slab_fuzz_exact_1000.log)"
grep '"error"\|"identical": false\|owned_sum_ok": false' $out/slab_fuzz_exact_1000.log | cut -c1-330 | head -5
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $out/pytest_gpu.log
