#!/bin/bash
# Developer sweep: regimes of the C3 cloud trajectory -> gpurun_out/sweep.log (one JSON line each)
out=gpurun_out/sweep.log; : > $out
for w in ${SWEEP_MARKS:-10 60 200}; do
  python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out || exit 1
done
