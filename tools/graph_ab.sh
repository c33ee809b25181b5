#!/bin/bash
# A/B of the hipGraph replay path (DESC_GRAPH=0 direct launches, 2 graph replays whatever the size) in one call
for w in C1 C2 C3 C4; do
  for g in 0 2; do
    DESC_GRAPH=$g python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-convergence > gpurun_out/graph_ab_${w}_$g.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open("gpurun_out/graph_ab_${w}_$g.json").read().strip().splitlines()[-1])
print("$w DESC_GRAPH=$g", "ms_per_step %.5f" % d["ms_per_step"], "kernel_ms %.5f" % d["roofline"]["kernel_ms"], "e2e %.2f / %.2f" % (d["end_to_end"]["ms"], d["end_to_end"]["repeat_ms"]))
PY
  done
done
