#!/bin/bash
# usage (GPU box): tools/ablate_prof.sh <workload> <masks...>  -- rocprofv3 kernel averages of the bench with the diagnostic libraries
wl=$1; shift
cd /tmp && export TMPDIR=/tmp
for n in 0 "$@"; do
  lib=$GRAFT_REPO_ROOT/desc_amd/libdesc_amd.so; [ $n != 0 ] && lib=$GRAFT_REPO_ROOT/tools/probes/libdesc_amd_abl$n.so
  export DESC_AMD_LIB=$lib
  rm -rf /tmp/abl_prof
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence > /dev/null 2>&1
  f=$(find /tmp/abl_prof -name "*kernel_stats.csv" | head -1)
  python3 - "$n" "$wl" "$f" <<'PY'
import csv, sys
n, wl, f = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "k_colsum_node" in r["Name"] or "k_sweep_band" in r["Name"]:
        out.append("%s avg %.1f us (%s calls)" % (r["Name"].split("(")[0].replace("void desc::", "").replace("desc::", ""), float(r["AverageNs"]) / 1e3, r["Calls"]))
print("ablate %s %s: %s" % (n, wl, "; ".join(out)))
PY
done
