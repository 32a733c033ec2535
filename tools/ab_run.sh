#!/bin/bash
# Developer: probe each variant library of tools/ab at the given warm-up points. usage: ab_run.sh "old a b" "10 400"
for w in $2; do for v in $1; do
  WSFLUID_LIBRARY=$PWD/tools/ab/lib$v.so python3 tools/probe.py c3 cloud $w 20 2>/dev/null | grep '^{' | sed "s/^{/{\"lib\": \"$v\", /"
done; done
