#!/bin/bash
# CEMP kernels with the per-sample arrays in uncached memory (default) vs ordinary (DESC_DEBUG_UNCACHED=6: bit 64 off)
cd /tmp && export TMPDIR=/tmp
for u in 70 6 70 6; do
  export DESC_DEBUG_UNCACHED=$u
  rm -rf /tmp/cprof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cprof -- python3 $GRAFT_REPO_ROOT/tools/cemp_probe.py ${1:-C4} > /dev/null 2>&1
  f=$(find /tmp/cprof -name "*kernel_stats.csv" | head -1)
  python3 - "$u" "$f" <<'PY'
import csv, sys
u, f = sys.argv[1:3]
out = []
for r in csv.DictReader(open(f)):
    if "cemp" in r["Name"]:
        out.append("%s avg %.1f us x %s" % (r["Name"].split("(")[0].split("::")[-1], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("uncached mask %s: %s" % (u, "; ".join(out)))
PY
done
