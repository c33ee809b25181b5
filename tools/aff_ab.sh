#!/bin/bash
# A/B of the band-affinity piece scheduling (DESC_DEBUG_AFFINITY = slack in K cycles; 0 = production list scheduling)
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for a in $AFF_LIST; do
    export DESC_DEBUG_AFFINITY=$a
    rm -rf /tmp/ab_prof
    DESC_DEBUG_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2> /tmp/ab_err.txt
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    pieces=$(grep -m1 "band sweep:" /tmp/ab_err.txt | sed 's/.*band sweep: //')
    python3 - "$a" "$wl" "$f" "$pieces" <<'PY'
import csv, sys
a, wl, f, pieces = sys.argv[1:5]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum_node" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3))
print("%s affinity=%-4s %s | %s" % (wl, a, "; ".join(out), pieces))
PY
  done
done
