#!/bin/bash
# usage (GPU box): tools/env_ab.sh "<ENV=VAL ...>" <workloads...> -- rocprofv3 kernel averages with and without the environment setting, alternating
envs=$1; shift
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  for mode in base env base env; do
    rm -rf /tmp/ab_prof
    if [ $mode = env ]; then export $envs; else for kv in $envs; do unset ${kv%%=*}; done; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2>&1
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    python3 - "$mode" "$wl" "$f" <<'PY'
import csv, sys
n, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum_node" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us (%s calls)" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("%-5s %s: %s" % (n, wl, "; ".join(out)))
PY
  done
done
