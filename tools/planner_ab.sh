#!/bin/bash
# A/B of one environment switch of the planner (AB_VAR, default DESC_DEBUG_PIECE_COST) over the values in PC_LIST: rocprofv3 kernel averages
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for a in $PC_LIST; do
    export ${AB_VAR:-DESC_DEBUG_PIECE_COST}=$a
    rm -rf /tmp/ab_prof
    DESC_DEBUG_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2> /tmp/ab_err.txt
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    python3 - "$a" "$wl" "$f" <<'PY'
import csv, sys
a, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3))
print("%s value=%-5s %s" % (wl, a, "; ".join(out)))
PY
  done
done
