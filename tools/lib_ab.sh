#!/bin/bash
# usage (GPU box): tools/lib_ab.sh <other .so> "<ENV=VAL ...|->" <workloads...> -- production library vs another build (with an optional environment for both), alternating
other=$1; envs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
[ "$envs" != "-" ] && export $envs
for wl in "$@"; do
  for lib in $GRAFT_REPO_ROOT/desc_amd/libdesc_amd.so $GRAFT_REPO_ROOT/$other $GRAFT_REPO_ROOT/desc_amd/libdesc_amd.so $GRAFT_REPO_ROOT/$other; do
    export DESC_AMD_LIB=$lib
    rm -rf /tmp/ab_prof
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence $BENCH_ARGS > /dev/null 2>&1
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    python3 - "$(basename $lib)" "$wl" "$f" <<'PY'
import csv, sys
n, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us (%s calls)" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("%-22s %s: %s" % (n, wl, "; ".join(out)))
PY
  done
done
