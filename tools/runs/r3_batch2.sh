#!/bin/bash
out=gpurun_out/r3b2; mkdir -p $out
python -m pytest tests/test_gpu_graph.py tests/test_gpu_host.py -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log
python3 tools/ablate.py 10 base full > $out/ablate10.log 2>&1
python3 tools/ablate.py 400 base full > $out/ablate400.log 2>&1
python3 tools/ablate.py 60 base full > $out/ablate60.log 2>&1
cat $out/ablate10.log $out/ablate60.log $out/ablate400.log
