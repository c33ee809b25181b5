#!/bin/bash
# usage (GPU box): LIBS="desc_amd/libdesc_amd.so tools/probes/libdesc_amd_exp1.so ..." tools/libs_ab.sh <workloads...> -- rocprofv3 kernel averages of
# bench.py with each library in turn (one box, one call); ENVS="A=1 B=2" is exported for all of them
cd /tmp && export TMPDIR=/tmp
[ -n "$ENVS" ] && export $ENVS
for wl in "$@"; do
  for lib in $LIBS; do
    export DESC_AMD_LIB=$GRAFT_REPO_ROOT/$lib
    rm -rf /tmp/ab_prof
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps ${STEPS:-30} --warmup ${WARMUP:-5} --no-cpu-baseline --no-convergence > /dev/null 2>&1
    f=$(find /tmp/ab_prof -name "*kernel_stats.csv" | head -1)
    python3 - "$(basename $lib)" "$wl" "$f" <<'PY'
import csv, sys
n, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3))
print("%-24s %s: %s" % (n, wl, "; ".join(out)))
PY
  done
done
