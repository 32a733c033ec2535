#!/bin/bash
# Developer: FETCH_SIZE / WRITE_SIZE per kernel for a variant library. usage: ab_traffic.sh <lib> <warm> <steps>
v=$1; warm=$2; steps=$3
export TMPDIR=/tmp WSFLUID_LIBRARY=$PWD/tools/ab/lib$v.so
for c in FETCH_SIZE WRITE_SIZE; do
  d=$GRAFT_REPO_ROOT/gpurun_out/abt_${v}_${warm}_$c; rm -rf $d
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/probe.py c3 cloud $warm $steps > /dev/null 2>&1
  python3 - "$d" "$c" "$steps" "$v" "$warm" <<'PY'
import csv,glob,sys,collections
d,c,steps,v,warm=sys.argv[1],sys.argv[2],int(sys.argv[3]),sys.argv[4],sys.argv[5]
f=glob.glob(d+"/*/*_counter_collection.csv")[0]
per=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]==c: per[r["Kernel_Name"].split("(")[0].replace("void ","")].append(float(r["Counter_Value"])*1024)
print(v,warm,c,{k:round(sum(x[-steps:])/len(x[-steps:])/1e6,1) for k,x in per.items() if k.startswith("k_") and len(x)>=steps})
PY
done
