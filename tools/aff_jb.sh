#!/bin/bash
# band affinity x j-block width (DESC_DEBUG_JBLOCK; 0 = the planner's own choice)
cd /tmp && export TMPDIR=/tmp
wl=$1; shift
for cfg in "$@"; do
  a=${cfg%%:*}; jb=${cfg##*:}
  export DESC_DEBUG_AFFINITY=$a DESC_DEBUG_JBLOCK=$jb
  rm -rf /tmp/ab_prof
  DESC_DEBUG_TIMING=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2> /tmp/ab_err.txt
  f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
  pieces=$(grep -m1 "band sweep:" /tmp/ab_err.txt | sed 's/.*band sweep: //')
  python3 - "$a" "$jb" "$wl" "$f" "$pieces" <<'PY'
import csv, sys
a, jb, wl, f, pieces = sys.argv[1:6]
out = []
for r in csv.DictReader(open(f)):
    if "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void desc::", ""), float(r["AverageNs"]) / 1e3))
print("%s affinity=%-3s jblock=%-4s %s | %s" % (wl, a, jb, "; ".join(out), pieces))
PY
done
