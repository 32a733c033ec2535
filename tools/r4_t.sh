cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 500 python3 tools/slab_fuzz.py 300 3 exact > $out/slab_fuzz_exact_300.log 2>&1; echo "exact 300: exit $? $(tail -1 $out/slab_fuzz_exact_300.log)"
timeout -k 10 500 python3 tools/slab_fuzz.py 100 5 lagged benign > $out/slab_fuzz_lagged_benign.log 2>&1; echo "lagged benign: exit $? $(tail -1 $out/slab_fuzz_lagged_benign.log)"
timeout -k 10 500 python3 tools/slab_fuzz.py 100 5 exact benign > $out/slab_fuzz_exact_benign.log 2>&1; echo "exact benign: exit $? $(tail -1 $out/slab_fuzz_exact_benign.log)"
grep '"error"' $out/slab_fuzz_lagged_benign.log $out/slab_fuzz_exact_300.log | cut -c1-330 | head -12
