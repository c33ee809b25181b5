#!/bin/bash
# usage (GPU box): [ENV=...] tools/wl_bench.sh <label> <workloads...>  -- one line per workload: kernel-pair time, ms/step, frac
label=$1; shift
for wl in "$@"; do
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-convergence 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s %s  kernel pair %.4f ms  ms/step %.4f  frac %.3f  e2e %.1f ms  err %.5f' % ('$label', '$wl', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['ms'], d['mean_abs_err_vs_truth']))"
done
