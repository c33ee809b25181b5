#!/bin/bash
# streamed arrays in uncached device memory (DESC_DEBUG_UNCACHED bit mask: 1 weights, 2 S0, 4 packed words) vs ordinary
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for u in $UC_LIST; do
    export DESC_DEBUG_UNCACHED=$u
    rm -rf /tmp/ab_prof
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2>&1
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    python3 - "$u" "$wl" "$f" <<'PY'
import csv, sys
u, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum_node" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3))
print("%s uncached=%s %s" % (wl, u, "; ".join(out)))
PY
  done
done
