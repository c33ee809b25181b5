#!/bin/bash
# what bounds the reference's own demo size (C1, n = 200): per-variant timing and the kernel durations
for v in 0 1 2 3; do
  DESC_DEBUG_VARIANT=$v python bench.py --workload C1 --steps 200 --warmup 20 --no-cpu-baseline --no-convergence > gpurun_out/c1_v$v.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/c1_v$v.json").read().strip().splitlines()[-1])
print("C1 DESC_DEBUG_VARIANT=$v", d["roofline"]["kernel"], "ms_per_step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
DESC_DEBUG_VARIANT=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1_prof_v$v -o c1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C1 --steps 200 --warmup 20 --no-cpu-baseline --no-convergence > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/c1_prof_v$v -name "*kernel_stats.csv" | head -1)
echo "== variant $v: $f"; head -8 "$f" | cut -c1-200
done
