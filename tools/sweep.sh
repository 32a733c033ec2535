#!/bin/bash
# Developer sweep: env-var variants x regimes -> gpurun_out/sweep.log (one JSON line each)
out=gpurun_out/sweep.log; : > $out
for u in 2 4 8; do
  for w in 10 200; do
    echo "# WS_UNROLL=$u warm=$w" >> $out
    WS_UNROLL=$u python tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' >> $out || exit 1
  done
done
