#!/bin/bash
# Developer (round 4): which read counters does gfx950 offer, and what do they say on kernels of known byte counts?
# Separate --pmc passes (the TCC block has 4 slots).  Output: gpurun_out/r04/calib/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r04/calib; mkdir -p $out
rocprofv3 -L > $out/list_avail.txt 2>&1
grep -o -E "\b(TCC|TCP)_[A-Z0-9_]+" $out/list_avail.txt | sort -u > $out/tcc_tcp_counters.txt
wc -l $out/tcc_tcp_counters.txt
tools/bin/traffic_calib > $out/expected.json
i=0
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_MISS_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); d=$out/pass$i; rm -rf $d
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- tools/bin/traffic_calib > $out/pass$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(out + "/pass*/*/*_counter_collection.csv"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        per[r["Counter_Name"]][r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, ks in per.items():
        for k, v in ks.items():
            v.sort()
            res[c][k] = [x for _, x in v]
json.dump(res, open(out + "/counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps({c: {k: v[-3:] for k, v in ks.items()} for c, ks in res.items()}, indent=1)[:6000])
PY
