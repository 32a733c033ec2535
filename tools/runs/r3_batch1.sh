#!/bin/bash
# round-3 developer batch: graph tests, K5 / K4 A/B, graph vs direct bench
out=gpurun_out/r3b1; mkdir -p $out
python -m pytest tests/test_gpu_graph.py tests/test_gpu_overflow.py tests/test_gpu_parity.py -q -m gpu > $out/tests.log 2>&1; tail -5 $out/tests.log
python3 tools/ablate.py 400 base pre nodens nomask > $out/ablate400.log 2>&1
python3 tools/ablate.py 10 base pre nodens nomask > $out/ablate10.log 2>&1
cat $out/ablate400.log $out/ablate10.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ieee --no-readback > $out/bench_direct.json 2> $out/bench_direct.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ieee --no-readback --graph > $out/bench_graph.json 2> $out/bench_graph.err
WS_BENCH_FORCE_SLAB=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ieee --no-readback > $out/bench_slab_direct.json 2> $out/bench_slab_direct.err
WS_BENCH_FORCE_SLAB=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ieee --no-readback --graph > $out/bench_slab_graph.json 2> $out/bench_slab_graph.err
for f in direct graph slab_direct slab_graph; do python3 -c "
import json,sys
d=json.load(open('$out/bench_$f.json'))
print('$f', round(d['ms_per_step'],4), [round(x,4) for x in d['repetitions']['ms_per_step']], 'settled', round(d['settled']['ms_per_step'],4), d.get('stats'))
"; done
