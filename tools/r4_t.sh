cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r04; mkdir -p $out
timeout -k 10 400 python3 tools/migration_peak.py c4 1 4 450 > $out/migration_peak_c4_exact.log 2>&1; echo "c4 exit $?"
timeout -k 10 700 python3 tools/migration_peak.py c5 1 8 450 > $out/migration_peak_c5_exact.log 2>&1; echo "c5 exit $?"
timeout -k 10 400 python3 tools/migration_peak.py c3 2 2 450 > $out/migration_peak_c3x2_exact.log 2>&1; echo "c3x2 exit $?"
python3 - <<'P'
import json
for c in ("c4","c5","c3x2"):
    rows=[json.loads(l) for l in open("gpurun_out/r04/migration_peak_%s_exact.log"%c) if l.startswith("{")]
    print(c, [("FAILED" in r) for r in rows].count(True), "failed;", [(r["stats"]["migration_now"], r["stats"]["halo_now"], r["stats"]["far_now"]) for r in rows if "stats" in r])
P
