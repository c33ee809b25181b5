#!/bin/bash
# rocprofv3 kernel averages of the sharded code path on one rank: direct (world 1) vs the multi-rank path forced (exchange layout, RCCL calls, unpack)
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  rm -rf /tmp/fprof
  DESC_FORCE_SHARDED=1 DESC_DEBUG_FORCE_COLLECTIVES=$f timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fprof -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${1:-C4} --steps 30 --warmup 5 > /tmp/fprof.json 2>/dev/null
  f2=$(find /tmp/fprof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$f2" <<'PY'
import csv, sys, json
forced, f = sys.argv[1:3]
d = json.loads(open("/tmp/fprof.json").read().strip().splitlines()[-1])
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("k_colsum_node", "k_sweep_band", "k_unpack_S", "ncclDevKernel", "copyBuffer", "rccl")):
        out.append("%s avg %.1f us x %s" % (n.split("(")[0].replace("void desc::", "").replace("desc::", "")[:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
print("forced=%s ms_per_step %.4f | %s" % (forced, d["ms_per_step"], "; ".join(out)))
PY
done
