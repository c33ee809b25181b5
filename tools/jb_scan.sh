#!/bin/bash
# usage (GPU box): tools/jb_scan.sh <workload> <JB values...>  -- sweep average for forced j-block widths (0 = the library's choice)
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
for jb in 0 "$@" 0; do
  export DESC_DEBUG_JBLOCK=$jb
  rm -rf /tmp/ab_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2>&1
  f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
  python3 - "$jb" "$wl" "$f" <<'PY'
import csv, sys
n, wl, f = sys.argv[1:4]
for r in csv.DictReader(open(f)):
    if "k_sweep_band" in r["Name"]:
        print("JB %4s %s: sweep avg %.1f us" % (n, wl, float(r["AverageNs"]) / 1e3))
PY
done
